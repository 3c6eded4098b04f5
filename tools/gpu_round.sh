cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_cli_resume.py -m gpu -x -q 2>&1 | grep -v "^[0-9-]*\.[0-9]* [0-9-]*\.[0-9]* [0-9-]*\.[0-9]* $" | tail -30
