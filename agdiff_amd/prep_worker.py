"""Worker process of agdiff_amd.driver.run_job: packs the next batch and builds its BatchTopology (host work only: numpy, the
library's host helpers; nothing here touches a GPU) while the parent samples the current one.

Started as `python -m agdiff_amd.prep_worker` -- a module of its own, NOT multiprocessing's spawn: that re-imports the parent's
__main__ in the child, i.e. re-runs any user script that lacks an `if __name__ == "__main__"` guard.  Protocol on stdin / stdout:
8-byte little-endian length + pickle.  Request: (bmols, confs, rank, world, topo_opts) or None (quit); reply: ("ok", (packed,
topology)) or ("error", text)."""
import os
import pickle
import struct
import sys


def _read(f):
    head = f.read(8)
    if len(head) < 8:
        return None
    n = struct.unpack("<q", head)[0]
    buf = bytearray()
    while len(buf) < n:
        chunk = f.read(min(n - len(buf), 1 << 24))
        if not chunk:
            return None
        buf += chunk
    return pickle.loads(bytes(buf))


def _write(f, obj):
    blob = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    f.write(struct.pack("<q", len(blob)))
    f.write(blob)
    f.flush()


def main():
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr                     # (anything printed by imports must not corrupt the reply stream)
    from agdiff_amd import driver
    import agdiff_amd.topology  # noqa: F401
    _write(out, ("ready", os.getpid()))
    while True:
        req = _read(inp)
        if req is None:
            return
        try:
            _write(out, ("ok", driver._prepare_in_worker(*req)))
        except Exception as e:      # (the parent prepares the batch itself)
            _write(out, ("error", "%s: %s" % (type(e).__name__, e)))


class Client:
    """Parent side: one request in flight at a time.  submit() sends and returns at once; result() blocks for the reply."""

    def __init__(self):
        import subprocess
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        self.proc = subprocess.Popen([sys.executable, "-m", "agdiff_amd.prep_worker"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                     env=env, cwd=root)
        self.ready = False
        self.pending = 0

    def _await_ready(self):
        if not self.ready:
            msg = _read(self.proc.stdout)
            if not msg or msg[0] != "ready":
                raise RuntimeError("the preparation worker did not start")
            self.ready = True

    def submit(self, *req):
        _write(self.proc.stdin, req)
        self.pending += 1

    def result(self):
        self._await_ready()
        msg = _read(self.proc.stdout)
        self.pending -= 1
        if msg is None:
            raise RuntimeError("the preparation worker went away (exit code %s)" % self.proc.poll())
        if msg[0] != "ok":
            raise RuntimeError("the preparation worker failed: %s" % (msg[1],))
        return msg[1]

    def close(self):
        try:
            if self.proc.poll() is None:
                try:
                    _write(self.proc.stdin, None)
                    self.proc.stdin.close()
                except Exception:
                    pass
                try:
                    self.proc.wait(timeout=5)
                except Exception:
                    self.proc.kill()
        finally:
            for f in (self.proc.stdout, self.proc.stdin):
                try:
                    f.close()
                except Exception:
                    pass


if __name__ == "__main__":
    main()
