import sys, torch
sys.path.insert(0, "/root/repo")
import bench
for geom, B in (("G128", 256), ("G256", 200)):
    rs = [bench.Runner(geom, B, seed=5 + 11 * k, steps_per_graph=4) for k in range(2)]
    want = []
    for r in rs:
        out = r.model.forward(r.z, r.x); torch.cuda.synchronize()
        want.append({k: getattr(out, k).clone() for k in ("score_map", "size_map", "offset_map", "pred_boxes", "hann_boxes", "conf")})
    bad = 0
    for rep in range(40):
        for i in range(50):
            for r in rs:
                r.graph_s.launch(r.stream)
        torch.cuda.synchronize()
        for r, w in zip(rs, want):
            for o in r.outs:
                for k, v in w.items():
                    if not torch.equal(getattr(o, k), v): bad += 1
    print(geom, "4000 interleaved graph launches (16000 steps) per shard, mismatches:", bad)
    for r in rs: r.close()
