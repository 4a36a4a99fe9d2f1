"""Per-phase cycle sums of render_fwd2_kernel over the busy tiles (diagnostic build: tools/build_render_probe2.sh)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import kernel_cases as kc
from hifihr_amd._lib import HifihrLib
from hifihr_amd.mano_tables import synthetic_mano_tables
lib = HifihrLib(os.path.join(R, "tools", "_probe", "libhifihr_render_stamp2.so"))
t = synthetic_mano_tables(0); B, H, aa, V = 32, 224, 3, 778
verts, vcol, cam, lc, ld = (x.cuda().contiguous() for x in kc.make_render_inputs(t, B, 7, H))
h = lib.renderer_create(t.faces, V, image_size=H, aa=aa)
ws = torch.zeros(lib.render_workspace_bytes(h, B), dtype=torch.uint8, device="cuda")
rgba = torch.empty(B, 4, H, H, device="cuda"); fid = torch.empty(B, H * aa, H * aa, dtype=torch.int32, device="cuda")
buf = (ctypes.c_ulonglong * 16)()
for _ in range(2):
    lib.render_fwd(h, verts, vcol, cam, lc, ld, rgba, fid, ws)
torch.cuda.synchronize()
hist = (ctypes.c_uint * 32)()
tm = (ctypes.c_ulonglong * 16)()
lib.c.hifihr_debug_render_stamps(buf, 1); lib.c.hifihr_debug_render_hist(hist, 1); lib.c.hifihr_debug_render_times(tm, 1)
lib.render_fwd(h, verts, vcol, cam, lc, ld, rgba, fid, ws)
torch.cuda.synchronize()
lib.c.hifihr_debug_render_stamps(buf, 0)
v = list(buf)
n = max(v[15], 1)
names = ["init (sxs, zbuf) + barrier", "stage faces (global -> LDS)", "raster pass (all)", "resolve + compaction", "shading", "  pass: rect + scan", "  pass: stage A rounds", "  pass: stage B (final drain)"]
print(f"busy tiles {v[15]}; per busy tile: faces {v[13] / n:.1f}, candidates {v[12] / n:.1f}, survivors (final drains) {v[11] / n:.1f}, busy pixels {v[14] / n:.1f}")
for i, nm in enumerate(names):
    print(f"  {nm:34s} {v[i] / n:9.0f} cycles = {v[i] / n / 2100:6.2f} us")
# evidence check (round 5 filed a stamps file whose first three phases and both counters read 0): the phases must account
# for the items' own durations, and the counters must be alive
top = sum(v[i] for i in range(5))
if v[15] == 0 or v[12] == 0 or v[11] == 0 or any(v[i] == 0 for i in range(8)) or (v[9] > 0 and top < 0.5 * v[9]):
    print(f"STAMPS INCOMPLETE: phases sum {top / n:.0f} cycles per tile against {v[9] / n:.0f} measured per resolving item; raw {v}")
    sys.exit(1)
print(f"  phases above (0-4) sum to {top / n:.0f} cycles per busy tile; the items' own clock: {v[9] / n:.0f} (a split tile's resolving part only)")
lib.c.hifihr_debug_render_hist(hist, 0)
print(f"slowest covered tile: {v[10] / 2100:.1f} us; covered tiles by duration (7.8 us buckets):", [int(x) for x in hist][:24])
lib.c.hifihr_debug_render_times(tm, 0)
tv = list(tm)
if tv[10] > 0:                                     # render_fwd3_kernel ran: the launch's timeline on the chip-wide 100 MHz clock
    t0 = tv[0]
    print(f"third form: {tv[10]} items ({tv[11]} of them parts of split tiles); relative to the first workgroup's start (us):")
    for c, nm in enumerate((">= 96 faces per part", "48-95", "24-47", "< 24")):
        if tv[1 + c]:
            print(f"  class {c} ({nm:20s}): last item STARTED at {(tv[5 + c] - t0) / 100:7.1f}, last item ENDED at {(tv[1 + c] - t0) / 100:7.1f}")
    print(f"  last workgroup (background strips included) ended at {(tv[9] - t0) / 100:7.1f}")
