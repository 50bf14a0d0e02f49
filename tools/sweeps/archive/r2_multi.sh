cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "many_queries or config5_every or config4_whole or full_size_databases or device_built or empty_records" 2>&1 | tail -25
timeout -k 10 300 python -m pytest tests/test_cli.py -m gpu -x -q 2>&1 | tail -5
