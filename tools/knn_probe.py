import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
mh = importlib.import_module("multi-h_amd")
sc = mh.synth.make_scene(50000, 10, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
for grid in (1, 0, 1, 0):
    e.set_tuning(31, grid)
    e.build_neighbors_knn(16, radius=200.0)
    t = time.perf_counter()
    for _ in range(5): e.build_neighbors_knn(16, radius=200.0)
    e.synchronize()
    print("grid" if grid else "exhaustive", (time.perf_counter() - t) / 5 * 1e3, "ms per neighbourhood build")
e.close()
