cd "$GRAFT_REPO_ROOT"
export LDW_AMD_LIB=$PWD/ldweaver_amd/libldweaver_amd_exp.so
run() { echo -n "$1: "; env $2 timeout -k 10 120 python tools/fuzz_paths.py --cases 100 --seed 203 --only 96 2>&1 | grep "^case" | sed 's/.*violations \([0-9]*\).*/violations \1/'; }
run base ""
run no_prune LDW_NO_PRUNE=1
run no_tab11 LDW_NO_TAB11=1
run no_fuse_tab LDW_NO_FUSE_TAB=1
run no_maybe LDW_NO_MAYBE=1
run r02_bound LDW_SCREEN_R02_BOUND=1
run no_span LDW_NO_SPAN=1
