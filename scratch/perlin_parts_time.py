"""Normalised Perlin call at batch 512, part by part (direct C-ABI calls): the look-ahead launch with and without its lattice job and its
statistics job, and the plain launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg, bench
pkg = sonar_pkg.load(); hl = pkg.hip_lib; lib = hl.load()
b, c, h, w = 512, 4, 128, 128
chw = c * h * w
st = hl._stream()
t0 = hl.perlin_lattice(2, c, h, w, "cuda", "lerp", 9, 101)
t1 = hl.perlin_lattice(2, c, h, w, "cuda", "lerp", 9, 103)
tout = torch.empty_like(t1)
out = torch.empty((b, c, h, w), device="cuda")
p0, p1 = hl.new_partials("cuda"), hl.new_partials("cuda")
assert lib.sonar_perlin_noise_ahead_f32(t0.data_ptr(), out.data_ptr(), b, chw, 2.0, 9, 100, 0, 0.9, 2.5, p0.data_ptr(), 0, 102, t1.data_ptr(), p1.data_ptr(), None, 0, 0, 0, 0, 0, 0, st) == 0
def ahead(next_, lat):
    return lambda: lib.sonar_perlin_noise_ahead_f32(t0.data_ptr(), out.data_ptr(), b, chw, 2.0, 9, 100, 0, 0.9, 2.5, p0.data_ptr(), 1, 102,
                                                    t1.data_ptr() if next_ else None, p1.data_ptr() if next_ else None, tout.data_ptr() if lat else None, 2 if lat else 0, c, h, w, 0, 103, st)
for name, fn in (("final pass alone (no statistics, no lattice)", ahead(False, False)), ("+ next call's statistics", ahead(True, False)),
                 ("+ statistics + a later call's lattice", ahead(True, True)), ("final pass + lattice", ahead(False, True)),
                 ("lattice launch alone", lambda: hl.perlin_lattice(2, c, h, w, "cuda", "lerp", 9, 105)),
                 ("un-normalised generate (terms given)", lambda: hl.perlin_generate((b, c, h, w), t0, 2.0, 9, 100, 0, partials=None)),
                 ("uniform fill of the same tensor", lambda: hl.philox_uniform((b, c, h, w), "cuda", 1, 7))):
    ts = sorted(bench.event_us(fn, 50, 10) for _ in range(5))
    print(f"{name:48s} {ts[2]:6.1f} us (min {ts[0]:.1f})", flush=True)
