set -u
cd /root/repo
export ROUND=r06
bash tools/gpu_session.sh tests bench benchlong prof refstyle matrix 2>&1 | tail -150
python tools/profile_small_calls.py > gpurun_out/small_calls.json 2> gpurun_out/small_calls.err
python tools/diag_small_call_passes.py 1e6 > gpurun_out/diag_passes_1e6.json 2>/dev/null
ab_bin/host_launch_cost pi-quant_amd/piquant/libpiquant.so > gpurun_out/host_launch_cost.txt 2>&1
(cd ab_old && python tools/dtype_matrix.py > ../gpurun_out/dtype_matrix_r05.json 2> ../gpurun_out/dtype_matrix_r05.err)
python tools/dtype_matrix.py > gpurun_out/dtype_matrix_again.json 2>/dev/null
bash tools/ab_bench.sh 3 ab_old . .,PIQUANT_HIP_REFERENCE_LAYOUT=0 > gpurun_out/ab_headline_library.txt 2>&1
cat gpurun_out/ab_headline_library.txt
