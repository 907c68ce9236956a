import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svit_amd import hip, ops
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblds_canary.so"))
lib.lds_canary_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
DEV = "cuda"
M = 13064
a16 = (torch.randn(M, 384, device=DEV) * 0.5).bfloat16()
w16 = (torch.randn(1536, 384, device=DEV) * 0.1).bfloat16()
tn_out = torch.zeros(384, 384, device=DEV)
nt_out = torch.empty((M, 1536), device=DEV, dtype=torch.bfloat16)
qa = (torch.randn(8, 4, 1633, 128, device=DEV) * 0.3).bfloat16()
ka = (torch.randn(8, 4, 457, 128, device=DEV) * 0.3).bfloat16()
vv = (torch.randn(8, 4, 457, 96, device=DEV) * 0.3).bfloat16()
def partner(kind):
    if kind == "tn": ops.gemm_tn(a16, a16, tn_out)
    elif kind == "nt": ops.gemm_nt(a16, w16, None, hip.EPI_BF16, out=nt_out)
    elif kind == "attn": ops.attn_fwd(qa, ka, vv, 96 ** -0.5)
side = torch.cuda.Stream()
for bytes_, threads in ((45888, 192), (42240, 192), (65536, 256), (16384, 256)):
    for pk in ("none", "tn", "nt", "attn"):
        blocks = 1024
        out = torch.zeros(blocks * threads * 4, dtype=torch.int32, device=DEV)
        tot = 0; ev = []
        for rep in range(6):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(8): partner(pk)
            rc = lib.lds_canary_launch(blocks, threads, bytes_, 200, out.data_ptr(), C.c_void_p(torch._C._cuda_getCurrentRawStream(0)))
            assert rc == 0, rc
            main.wait_stream(side)
            torch.cuda.synchronize()
            o = out.view(blocks, threads, 4)
            bad = o[:, :, 0].long().sum().item()
            tot += bad
            if bad and len(ev) < 3:
                idx = (o[:, :, 0] > 0).nonzero()
                b, t = idx[0].tolist()
                offs = sorted(set((o[:, :, 1][o[:, :, 0] > 0] * 4).tolist()))[:24]
                ev.append("rep %d: %d bad words in %d threads of %d blocks; first block %d thread %d: byte off %d val 0x%08x iter %d; byte offsets %s"
                          % (rep, bad, len(idx), len(set(idx[:, 0].tolist())), b, t, o[b, t, 1].item() * 4, o[b, t, 2].item() & 0xffffffff, o[b, t, 3].item(), offs))
        print("canary lds=%6d B x %d thr, partner %-5s: %d corrupted words" % (bytes_, threads, pk, tot), flush=True)
        for e in ev: print("     ", e)
