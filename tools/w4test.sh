set -e
mkdir -p gpurun_out/w4
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "winograd4 or benchmark_dispatch" > gpurun_out/w4/ops.log 2>&1 || { tail -30 gpurun_out/w4/ops.log; exit 1; }
tail -3 gpurun_out/w4/ops.log
B=64 N=10 python tools/time_wino4.py > gpurun_out/w4/time_new.log 2>&1
ONET_HIP_LIB=$PWD/onet_amd/libonet_hip_w4v1.so B=64 N=10 python tools/time_wino4.py > gpurun_out/w4/time_old.log 2>&1 || true
cat gpurun_out/w4/time_new.log gpurun_out/w4/time_old.log
