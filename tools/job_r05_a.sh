set -x
mkdir -p gpurun_out/r05
timeout 300 tools/ubench_valu_ceiling > gpurun_out/r05/valu_ceiling.json 2> gpurun_out/r05/valu_ceiling.err
cat gpurun_out/r05/valu_ceiling.json
cd /tmp && export TMPDIR=/tmp
for m in 3 4 5; do
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace -d /root/repo/gpurun_out/r05/vc_pmc_$m -o vc -- /root/repo/tools/ubench_valu_ceiling $m > /dev/null 2>&1
done
cd /root/repo
ls -R gpurun_out/r05 | head -30
timeout 1400 python -m pytest tests/test_gpu_production_parity.py -q 2>&1 | tail -30
