set -e
python -m pytest tests/test_gpu_conv.py -x -q > gpurun_out/c3_conv_tests.log 2>&1 || { tail -20 gpurun_out/c3_conv_tests.log; exit 1; }
tail -2 gpurun_out/c3_conv_tests.log
ALIBY_CONV_TALL64=1 python -m pytest tests/test_gpu_conv.py -x -q > gpurun_out/c3_conv_tests_tall.log 2>&1 || { tail -20 gpurun_out/c3_conv_tests_tall.log; exit 1; }
tail -2 gpurun_out/c3_conv_tests_tall.log
echo default; python3 scripts/bench_conv.py 288 plain
echo tall64; ALIBY_CONV_TALL64=1 python3 scripts/bench_conv.py 288 plain | grep "64->64"
echo tall64 B64T=2; ALIBY_CONV_TALL64=1 ALIBY_HIP_LIB=$PWD/aliby_amd/libaliby_hip_phase_C3_B64T_2.so python3 scripts/bench_conv.py 288 plain | grep "64->64"
echo tall64 B64T=4; ALIBY_CONV_TALL64=1 ALIBY_HIP_LIB=$PWD/aliby_amd/libaliby_hip_phase_C3_B64T_4.so python3 scripts/bench_conv.py 288 plain | grep "64->64"
echo B32T=2; ALIBY_HIP_LIB=$PWD/aliby_amd/libaliby_hip_phase_C3_B32T_2.so python3 scripts/bench_conv.py 288 plain | grep "32->32"
