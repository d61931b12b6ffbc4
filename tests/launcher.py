"""Process launcher of the GPU test session.

A test process that has initialised the GPU must not start programs itself on this pool (a fork of it still holds the
device, and an exec from there is refused).  So the session starts this helper FIRST, before anything touches the GPU
(conftest.pytest_configure), and tests ask it to run their child processes: compiled C-ABI clients, ranks of a sharded
batch.  The helper never imports torch and never touches the GPU.

Protocol (stdin / stdout, one JSON object per line):
  request  {"group": [{"argv": [...], "env": {...}} ...], "timeout": seconds, "cwd": path}
  reply    {"results": [{"rc": int, "out": str, "err": str} ...]}   (rc 124: stopped at the time limit)
The commands of a group run concurrently.
"""
import json
import os
import subprocess
import sys
import time


def run_group(req):
    procs = []
    for c in req["group"]:
        env = dict(os.environ)
        env.update(c.get("env") or {})
        procs.append(subprocess.Popen(c["argv"], env=env, cwd=req.get("cwd") or None, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE))
    deadline = time.time() + float(req.get("timeout", 600))
    results = [None] * len(procs)
    # (outputs are small: communicate() one after the other, all of them against the same deadline)
    for k, p in enumerate(procs):
        try:
            out, err = p.communicate(timeout=max(0.1, deadline - time.time()))
            results[k] = {"rc": p.returncode, "out": out.decode("utf-8", "replace"), "err": err.decode("utf-8", "replace")}
        except subprocess.TimeoutExpired:
            p.kill()   # exactly the process started above
            out, err = p.communicate()
            results[k] = {"rc": 124, "out": out.decode("utf-8", "replace"), "err": err.decode("utf-8", "replace")}
    return {"results": results}


def serve():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            reply = run_group(json.loads(line))
        except Exception as exc:   # noqa: BLE001
            reply = {"error": repr(exc)}
        sys.stdout.write(json.dumps(reply) + "\n")
        sys.stdout.flush()


class Launcher(object):
    """Client side (lives in the pytest process)."""

    def __init__(self):
        self.proc = subprocess.Popen([sys.executable, os.path.abspath(__file__)], stdin=subprocess.PIPE,
                                     stdout=subprocess.PIPE, bufsize=0)

    def run_group(self, group, timeout=600, cwd=None):
        req = {"group": group, "timeout": timeout, "cwd": cwd}
        self.proc.stdin.write((json.dumps(req) + "\n").encode())
        self.proc.stdin.flush()
        reply = json.loads(self.proc.stdout.readline().decode())
        if "error" in reply:
            raise RuntimeError("launcher: " + reply["error"])
        return reply["results"]

    def run(self, argv, env=None, timeout=600, cwd=None):
        return self.run_group([{"argv": argv, "env": env or {}}], timeout, cwd)[0]

    def close(self):
        if self.proc.poll() is None:
            self.proc.stdin.close()
            try:
                self.proc.wait(timeout=10)
            except subprocess.TimeoutExpired:
                self.proc.kill()


if __name__ == "__main__":
    serve()
