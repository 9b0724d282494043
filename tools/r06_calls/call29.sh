# the whole GPU suite + smoke on the final tree
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c29; mkdir -p $O
S=$SECONDS
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$? in $((SECONDS-S)) s"; tail -3 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
