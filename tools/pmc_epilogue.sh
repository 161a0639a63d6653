#!/bin/bash
# SQ counters of the plain path's fp64 epilogue kernels (k_mi_epilogue_fast / _rest, or k_mi_epilogue under LDW_NO_EPI_SPLIT) with the derived fractions
# bench.py quotes (roofline_mi_produced.epilogue): separate rocprofv3 --pmc passes over a short serial run of the plain path (tools/pmc_run.sh).
#   usage (GPU box): tools/pmc_epilogue.sh <out.json>
out=${1:-gpurun_out/pmc_epilogue.json}
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_run.sh "k_mi_epilogue" "$out" "--no-mixed --screen 0 --path 1" \
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD" \
  "SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_IFETCH" > "${out%.json}.log" 2>&1
python3 - "$out" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, e in d.items():
    if not isinstance(e, dict) or "SQ_BUSY_CU_CYCLES" not in e:
        continue
    cu = e["SQ_BUSY_CU_CYCLES"]
    e["derived"] = dict(valu_issue_frac=e["SQ_INSTS_VALU"] / cu, valu_busy_frac=e["SQ_ACTIVE_INST_VALU"] / cu, salu_per_valu=e["SQ_INSTS_SALU"] / e["SQ_INSTS_VALU"],
                        lanes_active_per_valu=e.get("SQ_THREAD_CYCLES_VALU", 0) / e["SQ_INSTS_VALU"], wait_inst_any_frac=e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"],
                        wait_any_frac=e.get("SQ_WAIT_ANY", 0) / e["SQ_WAVE_CYCLES"], waves=e["SQ_WAVES"], cu_busy_ms_at_2p4GHz=cu / 256 / 2.4e9 * 1e3)
d["_how"] = "tools/pmc_epilogue.sh: per-launch averages, bench.py --steps 1 --warmup 1 --no-overlap --no-mixed --screen 0 --path 1 (C4), two --pmc passes"
json.dump(d, open(sys.argv[1], "w"), indent=1)
for k, e in d.items():
    if isinstance(e, dict) and "derived" in e:
        print(k, {a: round(b, 4) for a, b in e["derived"].items()}, "VALU insts per launch", round(e["SQ_INSTS_VALU"]))
PY
