"""debug: sequences of sharded runs in ONE process (in-process communicator); every rank's error printed.
  python scripts/halo_p2p_debug.py <world>:<transport>:<halo> ...      e.g. 2:p2p:0 4:p2p:0"""
import os, sys, threading
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
os.environ["CUOPT_AMD_SHARD_DATAFLOW"] = "owner"
from cuopt_amd import capi, synthetic
p = synthetic.generate(262144, 262144, 10, seed=2, band=500)
def run(tag, world):
    cid = capi.softcomm_id(world)
    def worker(rank):
        try:
            s = capi.Solver(p, rank=rank, world=world, comm_id=cid, tol=0.0, iteration_limit=120)
            w = s.device.wire_bytes()
            r = s.advance()
            x, y, rc = s.solution()
            print(tag, "rank", rank, r["status_name"], r["steps_taken"], w["halo"], flush=True)
            s.close()
        except Exception as e:
            print(tag, "rank", rank, "ERROR", str(e)[:90], flush=True)
    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
for spec in sys.argv[1:]:
    world, transport, halo = spec.split(":")
    os.environ["CUOPT_AMD_SHARD_TRANSPORT"] = transport
    if halo == "0":
        os.environ["CUOPT_AMD_TUNE"] = "shard_halo=0"
    else:
        os.environ.pop("CUOPT_AMD_TUNE", None)
    run(spec, int(world))
