timeout 200 python tests/host_tools/tp70_proc_diag.py 2 16 2>&1 | tail -9
timeout 200 python tests/host_tools/tp70_proc_diag.py 2 0 2>&1 | tail -9
