"""A/B of the chunked LBS forward (DPOSER_LBS_FWD_CHUNK = 0: one blend GEMM + one skinning launch over the batch; n: chunks of n poses, the
skinning of chunk i on a side stream beside the blend GEMM of chunk i + 1), interleaved child processes, with a checksum of the vertices
(every mode must produce the same bits):   python tools/lbs_chunk_ab.py [--chunks 0,512,1024,2048]"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to("cuda:0")
    for n in (4096, 16384):
        pose = (torch.randn(n, 63, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(n)) * 0.3).contiguous()
        with torch.no_grad():
            for _ in range(3):
                out = bm(pose_body=pose)
            torch.cuda.synchronize()
            sha = hashlib.sha1(out.v.cpu().numpy().tobytes() + out.Jtr.cpu().numpy().tobytes()).hexdigest()[:12]
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    bm(pose_body=pose)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
        ts.sort()
        print(f"chunk={os.environ.get('DPOSER_LBS_FWD_CHUNK', '0'):>5} n={n:6d}  median {ts[2]:7.3f} ms  best {ts[0]:7.3f} ms  {n / ts[2] / 1e3:6.2f} M poses/s  sha1 {sha}", flush=True)
else:
    chunks = "0,512,1024,2048"
    if "--chunks" in sys.argv:
        chunks = sys.argv[sys.argv.index("--chunks") + 1]
    for rnd in range(2):
        for c in chunks.split(","):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DPOSER_LBS_FWD_CHUNK=c), check=True)
