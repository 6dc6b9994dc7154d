"""Dev tool: host and device timestamps of the staggered serving loop's stages for a few scenes (where does the
period go?).  Host times by perf_counter, device times by events recorded on the lanes, both relative to one origin."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene
dev = torch.device("cuda", 0)
ns = 4
batches = [bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234 + i)]), dev) for i in range(ns)]
model = bench.build_model(dev, probe_batch=batches[0])
lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
def ev(stream):
    e = torch.cuda.Event(enable_timing=True); e.record(stream); return e
log = []
def loop(n):
    stretch, head, prev = (), None, None
    for i in range(n):
        lane = lanes[i % 2]
        np.random.seed(1000 + i)
        rec = {"i": i, "h0": time.perf_counter()}
        with torch.cuda.stream(lane), torch.no_grad():
            for e in stretch: lane.wait_event(e)
            rec["d_gate"] = ev(lane)
            h = model.forward_split(batches[i % ns], 300, training=False, defer_proposals=True)
            rec["h_p1"] = time.perf_counter(); rec["d_bb"] = h.backbone_done
            h.advance()
            rec["h_adv"] = time.perf_counter(); rec["d_p1b"] = ev(lane)
        stretch = h.stretch_done
        rec["d_stretch"] = list(stretch)
        if head is not None:
            ph, pl = head
            with torch.cuda.stream(pl), torch.no_grad():
                rec["d_tail0"] = ev(pl)
                out = ph.finish()
                rec["d_tail1"] = ev(pl)
            if prev is not None and not isinstance(prev.get("proposal_scores"), (tuple, type(None))):
                prev["proposal_scores"] = prev["proposal_scores"].get()
            prev = out
        rec["h_tail"] = time.perf_counter()
        head = (h, lane)
        log.append(rec)
    ph, pl = head
    with torch.cuda.stream(pl), torch.no_grad(): ph.finish()
    torch.cuda.synchronize()
loop(6); log.clear()
torch.cuda.synchronize()
base_e = torch.cuda.Event(enable_timing=True); base_e.record(); torch.cuda.synchronize(); base_h = time.perf_counter()
# (timing events need enable_timing: the model's own events are not -- re-record stand-ins where needed)
loop(10)
def d(e):
    try: return base_e.elapsed_time(e)
    except Exception: return float("nan")
for r in log[3:8]:
    h = lambda k: (r[k] - base_h) * 1e3
    print("scene %d  host: start %.2f  part1 done %.2f  advance done %.2f  tail queued %.2f | device: gate passed %.2f  after P1b launches(SA end) %.2f  tail(i-1) %.2f .. %.2f" % (
        r["i"], h("h0"), h("h_p1"), h("h_adv"), h("h_tail"), d(r["d_gate"]), d(r["d_p1b"]), d(r.get("d_tail0", base_e)), d(r.get("d_tail1", base_e))))
