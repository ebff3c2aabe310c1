set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c_abi_exchange or packed_exchange or shift_exchange or overlap_streams" --timeout=200 > gpurun_out/pytest_xchg.log 2>&1
echo rc=$?; tail -30 gpurun_out/pytest_xchg.log
