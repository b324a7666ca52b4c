python tests/host_tools/tp_selftest_diag.py 2 4
python tests/host_tools/tp_selftest_diag.py 2 64
python tests/host_tools/tp_selftest_diag.py 2 64 0.5
python tests/host_tools/tp_selftest_diag.py 4 64
python tests/host_tools/tp_selftest_diag.py 2 1000
python tests/host_tools/tp_rank_time.py 4 8 2>&1 | tail -9
LLMK_Q4_KS=0 python tests/host_tools/tp_rank_time.py 4 8 2>&1 | tail -9
