./llm.f90_amd/csrc/probes/q4_mix_probe | tee gpurun_out/q4_mix_probe.jsonl
python tests/host_tools/ks_check.py
