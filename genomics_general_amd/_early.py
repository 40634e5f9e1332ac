"""The HIP runtime's start-up beside the interpreter's own.

The first HIP call of a process (hipGetDeviceCount: the runtime opens the driver, reads the topology, creates its queues) takes
0.09 - 0.13 s on an MI355X box (tools/ctx_time.py, profiles/r06/ctx_time_fresh_processes.txt) -- a third of a mid-size `.geno.gz`
run.  cli.Run already creates the device context on a helper thread beside the opening of the input; this module lets a driver
start the runtime's own start-up even earlier: the root scripts call start() before they import numpy and the package, so the
0.1 s run beside the imports, the argument parsing and the sample / window set-up.  Nothing here needs numpy; the library is
loaded by its path (the later _lib.lib() finds it loaded) and asked for the device count, whose answer is ignored (no device,
no library: the run itself says so where it needs them)."""
import ctypes
import os
import threading

_thread = None


def start():
    global _thread
    if _thread is not None or os.environ.get("PG_EARLY_INIT", "1") == "0":
        return _thread
    path = os.environ.get("PG_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpopgen_hip.so")

    def run():
        try:
            n = ctypes.c_int(0)
            ctypes.CDLL(path).pg_device_count(ctypes.byref(n))            # (ctypes releases the interpreter lock for the call)
        except (OSError, AttributeError):
            pass
    _thread = threading.Thread(target=run, daemon=True, name="hip-runtime-start")
    _thread.start()
    # a process that ends at once (-h, an argument error) must not tear the runtime down while this thread is still inside its start-up
    import atexit
    atexit.register(lambda: _thread.join(timeout=10.0))
    return _thread
