cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | grep -v "^[0-9-]*\.[0-9]* [0-9-]*\.[0-9]* [0-9-]*\.[0-9]* $" | tail -15
