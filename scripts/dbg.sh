cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x > gpurun_out/first_run.log 2>&1
echo rc=$?
tail -5 gpurun_out/first_run.log
grep -n -i "fault\|abort\|fatal" gpurun_out/first_run.log | head
