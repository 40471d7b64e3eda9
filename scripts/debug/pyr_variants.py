import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_amd")]
    import torch, droid_backends as db
    tm, waves, h, w, E = map(int, sys.argv[1:6])
    db.set_option("pyr_build_tm", tm); db.set_option("pyr_build_waves", waves)
    torch.manual_seed(0)
    f1 = torch.randn(E, 128, h, w, device="cuda").half(); f2 = torch.randn(E, 128, h, w, device="cuda").half()
    p = db.corr_pyramid_build(f1, f2); torch.cuda.synchronize()
    print("ok", tm, waves, h, w, E, float(p.float().abs().mean()))
else:
    for a in ((0, 8, 8, 16, 1), (0, 8, 16, 32, 1), (0, 4, 48, 64, 1), (0, 8, 48, 64, 1), (1, 8, 48, 64, 1), (1, 4, 48, 64, 1), (1, 8, 48, 64, 64)):
        r = subprocess.run(["timeout", "60", sys.executable, __file__] + [str(x) for x in a], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print(a, "rc", r.returncode, r.stdout.strip()[-80:], (r.stderr.strip().splitlines() or [""])[0][:100], flush=True)
