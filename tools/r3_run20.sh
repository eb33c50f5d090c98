cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest20.log 2>&1; tail -4 gpurun_out/r3_pytest20.log; grep -h "DDPM-250" gpurun_out/r3_pytest20.log
