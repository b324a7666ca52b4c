./llm.f90_amd/csrc/probes/q4_mix_probe | tee gpurun_out/q4_mix_probe.jsonl
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03g_pytest.log; cat gpurun_out/r03g_pytest.log
