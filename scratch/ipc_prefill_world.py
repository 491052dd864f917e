"""tests/test_tp.py's rank-process worker in its 'prefill' scenario (8.5 k-token prefills, the large all-reduces slot by slot through the hipIpc arenas: no RCCL with
several ranks on one device) at a world size the suite does not run it at: python scratch/ipc_prefill_world.py 4"""
import os, sys, time, pathlib, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_tp
world = int(sys.argv[1])
t0 = time.time()
got = test_tp._run_ipc_workers(pathlib.Path(tempfile.mkdtemp()), world, "float16", "prefill")
print(f"world {world}: {time.time() - t0:.0f} s; per rank [chunks(overlap 1), chunks(serial), overlapped == serial, steps, microbatches, two-microbatch == serial]:", [g[1] for g in got], flush=True)
