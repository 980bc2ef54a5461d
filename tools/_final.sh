set -u
mkdir -p gpurun_out/r5h
timeout 3000 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5h/pytest_full.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5h/pytest_full.log
