"""A/B of library builds on the headline workload (GPU box): python tools/ab.py libA.so libB.so ... [--rays N] [--brdf]
Each library runs in its own child process; prints trace / binning / total kernel ms (best of 5) and a histogram checksum."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("ISX_AB_CHILD"):
    sys.path.insert(0, ROOT)
    import zlib
    import altair_raytracing_amd as isx
    isx.load(); isx.init(0)
    for kv in filter(None, os.environ.get("ISX_AB_OPTS", "").split(",")):
        k, v = kv.split("="); isx.set_option(k, int(v))
    c = isx.default_config()
    if os.environ.get("ISX_AB_BRDF"):
        c.source_model = isx.SOURCE_BRDF; c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
    n = int(os.environ["ISX_AB_RAYS"])
    isx.fluxmap(c, 1_000_000, 1)
    best = None
    for _ in range(5):
        h, st = isx.fluxmap(c, n, 5)
        k = isx.last_kernel_ms()
        if best is None or st.t_kernel_ms < best[0]: best = (st.t_kernel_ms, k[1], k[2], k[0])
    print(json.dumps({"total_ms": round(best[0], 3), "trace_ms": round(best[1], 3), "bin_ms": round(best[2], 3), "single_ms": round(best[3], 3),
                      "Mrays_s": round(n / best[0] / 1e3, 1), "crc": zlib.crc32(h.tobytes()), "increments": st.bin_increments, "wall_hits": st.wall_hits}))
    sys.exit(0)
args = sys.argv[1:]
rays, brdf, libs = "50000000", "", []
while args:
    a = args.pop(0)
    if a == "--rays": rays = args.pop(0)
    elif a == "--brdf": brdf = "1"
    else: libs.append(a)
for lib in libs:
    path, _, opts = lib.partition(":")
    env = dict(os.environ, ISX_AB_CHILD="1", ISX_LIB_PATH=os.path.join(ROOT, path), ISX_AB_RAYS=str(int(float(rays))), ISX_AB_BRDF=brdf, ISX_AB_OPTS=opts)
    r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=900)
    print(lib, r.stdout.strip() or r.stderr[-600:], flush=True)
