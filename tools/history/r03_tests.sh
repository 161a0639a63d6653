#!/bin/bash
# whole GPU suite, verbose log under gpurun_out (progress is visible per test)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 ${TEST_TIMEOUT:-1100} python -m pytest tests -m gpu -x -v --durations=20 $PYTEST_ARGS > gpurun_out/r03_tests.log 2>&1
rc=$?
echo "pytest rc $rc"
tail -45 gpurun_out/r03_tests.log
exit $rc
