#!/bin/bash
# Diagnostic: run a command; if it is still alive after $1 seconds, dump every thread's native stack with rocgdb
# (the child allows any tracer: PR_SET_PTRACER_ANY) and the Python stacks (faulthandler), then kill it.
limit=$1; shift
out=${HANG_OUT:-gpurun_out/hang_probe}
python -X faulthandler -c "
import ctypes, faulthandler, runpy, sys
ctypes.CDLL(None).prctl(0x59616d61, ctypes.c_ulong(-1), 0, 0, 0)
faulthandler.dump_traceback_later($limit, exit=False, file=open('$out.py_stacks.txt', 'w'))
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')
" "$@" > $out.stdout 2> $out.stderr &
pid=$!
for ((i = 0; i < limit + 5; i++)); do
  sleep 1
  kill -0 $pid 2>/dev/null || { wait $pid; echo "finished rc=$? after ${i}s"; exit 0; }
done
echo "still running after $limit s: dumping stacks of $pid"
timeout 120 /opt/rocm/bin/rocgdb -p $pid -batch -ex "set pagination off" -ex "thread apply all bt 25" > $out.native_stacks.txt 2>&1
kill $pid; sleep 2; kill -9 $pid 2>/dev/null
echo "killed"
exit 3
