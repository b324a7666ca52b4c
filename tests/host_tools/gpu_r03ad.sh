mkdir -p gpurun_out/pf
timeout 1200 python -m pytest tests/test_prefill_gpu.py -x -q 2>&1 | tail -6
for env in "" "LLMK_PF_F32_MFMA=1"; do
  echo "== $env"
  env $env python bench.py --prefill 512 --type f16 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-700
done | tee gpurun_out/pf/f16_hm.txt
LLMK_WT=1 python tests/host_tools/pf_time.py 2>&1 | tail -15 | tee -a gpurun_out/pf/f16_hm.txt
LLMK_PF_F32_MFMA=1 LLMK_WT=1 python tests/host_tools/pf_time.py 2>&1 | tail -5 | tee -a gpurun_out/pf/f16_hm.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/pf/prof -- python $GRAFT_REPO_ROOT/bench.py --prefill 512 --type f16 --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/pf/prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -14 $f | cut -c1-200
