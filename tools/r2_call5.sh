cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_host.py tests/test_gpu_cli.py -q -x -p no:cacheprovider 2>&1 | tail -5
echo "== host path rates"
timeout -k 10 600 python tools/host_path_rate.py
