"""Diagnostic: LDS fill rate when every byte comes from the XCD's L2 (not L1, not MALL/HBM): each workgroup re-reads its own
4 K-steps x 28 KiB (3.7 MB per XCD) for many passes."""
import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfill_rate.so"))
lib.fill_rate2.argtypes = [C.c_int] * 3 + [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
K = 1536
cyc = torch.zeros(1, dtype=torch.int64, device="cuda")
sink = torch.zeros(4, device="cuda")
for wgs_per_cu in (1, 2):
    wgs = 256 * wgs_per_cu
    rows = 224 * wgs                                   # one private panel per workgroup
    A = torch.randn(rows, K, device="cuda").bfloat16()
    for ksteps, reps in ((4, 300), (24, 40)):
        for stages in (2, 3):
            for mode in (0, 1):
                if mode == 1 and stages != 2:
                    continue
                for _ in range(2):
                    rc = lib.fill_rate2(mode, 128, stages, A.data_ptr(), K, rows, ksteps, reps, wgs, cyc.data_ptr(), sink.data_ptr(), 0, 0)
                    torch.cuda.synchronize()
                assert rc == 0
                c = int(cyc.item())
                b = reps * ksteps * 28672
                print("%d WG/CU  %2d K-steps re-read (%5.1f MB per XCD)  %-9s %d stages: %6.1f B/clk per WG  %6.1f per CU" %
                      (wgs_per_cu, ksteps, wgs / 8 * ksteps * 28672 / 1e6, "LDS-DMA" if mode == 0 else "registers", stages,
                       b / c, wgs_per_cu * b / c), flush=True)
