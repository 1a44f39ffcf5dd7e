"""The instrumentation that must work without a GPU (include/bof_hip.h, "Instrumentation"): the event ring's dump entry
point and the $BOF_CRASH_TRACE signal handler.  No compute call is made; the library only has to load."""
import os
import signal
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "blas-on-flash_amd", "lib", "libbof_hip.so")


def test_event_dump_writes_a_header_even_before_any_call(tmp_path):
    out = tmp_path / "events.txt"
    code = (f"import ctypes; L = ctypes.CDLL({LIB!r}); L.bof_event_dump.restype = ctypes.c_uint64; "      # = events recorded so far
            f"raise SystemExit(int(L.bof_event_dump({str(out)!r}.encode())))")
    assert subprocess.run([sys.executable, "-c", code], timeout=120).returncode == 0
    text = out.read_text()
    assert text.startswith("[bof events] bof_event_dump: last 0 of 0 events"), text[:200]


def test_event_dump_to_a_path_that_cannot_be_opened_does_not_take_the_process_down():
    code = (f"import ctypes; L = ctypes.CDLL({LIB!r}); L.bof_event_dump.restype = ctypes.c_uint64; "
            "raise SystemExit(int(L.bof_event_dump(b'/nonexistent_dir/x/events.txt')))")
    assert subprocess.run([sys.executable, "-c", code], timeout=120).returncode == 0


def test_crash_trace_prints_the_native_stack_and_keeps_the_signal(tmp_path):
    """BOF_CRASH_TRACE=1: a fatal signal prints '[bof] fatal signal' + the stack + the ring, then the default action runs
    (the process still dies of the signal: the handler must not turn a crash into a clean exit)."""
    code = f"import ctypes, os, signal; ctypes.CDLL({LIB!r}); os.kill(os.getpid(), signal.SIGSEGV)"
    for on in ("1", "0"):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BOF_CRASH_TRACE=on), stderr=subprocess.PIPE,
                           text=True, timeout=120)
        assert p.returncode == -signal.SIGSEGV, (on, p.returncode, p.stderr[-500:])
        assert ("[bof] fatal signal" in p.stderr) == (on == "1"), p.stderr[-800:]
        if on == "1":
            assert "libbof_hip.so" in p.stderr and "[bof events] fatal signal" in p.stderr, p.stderr[-800:]
