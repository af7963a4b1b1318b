"""Throughput of the grow instances beyond the 64-cell row: the two-word fast kernels (Mask128) and the general grow instance (csrc/cape_grow_general.hip): frames resident in HBM, HIP-event timings of stage A and
stage B per call, (a) on the 640x480 / 1280x960 grids where CAPE_GROW=general can be compared with the fast kernels on the same
frames, (b) on grids only it serves (1920x1080, 1080x1920, 2560x1440).   python profiles/general_instance_rate.py [out.txt]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rgb-d-slam_amd", "python"))

import torch  # noqa: E402

from cape_amd import Extractor, synth, synth_gpu  # noqa: E402


def run(W, H, n, scene, cyl, general, reps=5):
    intr = {k: v * W / 640.0 for k, v in synth.DEFAULT_INTRINSICS.items()}
    dev = synth_gpu.stream(scene, 1, n, width=W, height=H)
    if general:
        os.environ["CAPE_GROW"] = "general"
    ex = Extractor(W, H, cylinders=cyl, max_batch=n, **intr)
    os.environ.pop("CAPE_GROW", None)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        ex.extract_device(dev.data_ptr(), n, st)
    torch.cuda.synchronize()
    ex.enable_timing(True)
    ex.reset_timings()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ex.extract_device(dev.data_ptr(), n, st)
    e1.record()
    torch.cuda.synchronize()
    t = ex.timings()
    ms = e0.elapsed_time(e1) / reps
    info = ex.spill_info()
    ex.close()
    return dict(ms=ms, fps=n / ms * 1e3, a_ms=t["cell_fit_s"] / reps * 1e3, b_ms=t["grow_s"] / reps * 1e3, general_frames=info[2])


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    rows = []
    for (W, H, n) in ((640, 480, 4096), (1280, 960, 1024)):
        for scene in ("room", "tunnel"):
            for cyl in (False, True):
                for general in (False, True):
                    rows.append((W, H, n, scene, cyl, general, run(W, H, n, scene, cyl, general)))
    for (W, H, n) in ((1920, 1080, 1024), (1080, 1920, 1024), (2560, 1440, 512)):
        for scene in ("room", "tunnel"):
            for cyl in (False, True):
                rows.append((W, H, n, scene, cyl, True, run(W, H, n, scene, cyl, False)))
    print(f"{'grid':>11} {'frames':>6} {'scene':>7} {'cyl':>4} {'instance':>9} {'ms/call':>9} {'frames/s':>11} {'stage A ms':>10} {'stage B ms':>10}", file=out)
    for W, H, n, scene, cyl, general, r in rows:
        print(f"{W:>6}x{H:<4} {n:>6} {scene:>7} {str(cyl):>4} {('general' if r['general_frames'] else ('fast-128' if W > 1280 else 'fast')):>9} {r['ms']:>9.3f} {r['fps']:>11.0f} "
              f"{r['a_ms']:>10.3f} {r['b_ms']:>10.3f}", file=out)


if __name__ == "__main__":
    main()
