import sys, os, itertools
sys.path.insert(0, "/root/repo")
import bench
for geom, B in (("G256", 64), ("G256", 128), ("G128", 64)):
    r = bench.Runner(geom, B, steps_per_graph=4)
    r.prewarm(0.3)
    st = r.stage_times(50)
    t = r.time_us(lambda: r.graph_s.launch(r.stream), 50) / r.S
    print(geom, B, {k: round(v, 1) for k, v in st.items()}, "step", round(t, 1), "frames/s", round(B / t * 1e6))
    r.close()
