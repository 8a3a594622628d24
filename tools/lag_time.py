import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import proof_systems_amd.khip as khip
khip.init(0)
for logn in (12, 16):
    srs = khip.Srs.create(khip.VESTA, 1 << logn)
    khip.sync(); t = time.perf_counter(); srs.compute_lagrange(logn); khip.sync()
    print("KH_LAG_QUAD=%s lagrange 2^%d: %.1f ms" % (os.environ.get("KH_LAG_QUAD", "1"), logn, 1e3 * (time.perf_counter() - t)))
    srs.close()
