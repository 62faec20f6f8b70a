"""One process per GPU, started by the script itself: `script.py --gpus N` spawns N fresh rank processes (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1) BEFORE the parent imports torch or makes any HIP call --
never a re-exec of a process that has touched the GPU.  Used by bench.py and tools/run_configs.py; this module imports neither
torch nor the HIP library.
"""
import os
import socket
import subprocess
import sys
import threading
import time


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(script, argv, n):
    """Run `python script argv...` as ranks 0..n-1; relay rank 0's stdout; return the first non-zero exit code (a rank that dies
    takes the others down: they would wait in the rendezvous for ever)."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else None))
    out0 = []
    rc = 0

    def pump():
        for raw in procs[0].stdout:
            out0.append(raw.decode())

    th = threading.Thread(target=pump, daemon=True)
    th.start()
    live = set(range(n))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for o in live:
                    procs[o].terminate()  # exactly the PIDs started above
        time.sleep(0.05)
    th.join(timeout=5)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    return rc
