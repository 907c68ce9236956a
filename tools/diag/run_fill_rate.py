"""Diagnostic: bytes per clock a CU pulls into LDS, by access pattern / mechanism / depth / residency."""
import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfill_rate.so"))
lib.fill_rate.argtypes = [C.c_int] * 3 + [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
rows, K = 13064 // 448 * 448, 1536          # a [12992 x 1536] bf16 operand (40 MB: MALL-resident when replayed)
A = torch.randn(rows, K, device="cuda").bfloat16()
Ac = A.clone()                               # the same bytes, read as contiguous 1-KiB pieces
cyc = torch.zeros(1, dtype=torch.int64, device="cuda")
sink = torch.zeros(4, device="cuda")
reps = 20
for wgs_per_cu in (1, 2):
    for mode, rb, stages in ((0, 64, 2), (0, 128, 2), (0, 256, 2), (0, 1024, 2), (0, 64, 3), (0, 128, 3), (0, 256, 3),
                             (0, 1024, 3), (0, 128, 4), (0, 1024, 4), (1, 128, 2), (1, 1024, 2)):
        ksteps = K * 2 // rb if rb < 1024 else 24
        ld = K if rb < 1024 else 512
        nrows = rows if rb < 1024 else rows * K // 512 - 56     # (the contiguous walk runs 23 KiB past a panel)
        src = A if rb < 1024 else Ac
        wgs = 256 * wgs_per_cu
        for _ in range(2):
            rc = lib.fill_rate(mode, rb, stages, src.data_ptr(), ld, nrows, ksteps, reps, wgs, cyc.data_ptr(), sink.data_ptr())
            torch.cuda.synchronize()
        assert rc == 0, rc
        c = int(cyc.item())
        bytes_per_wg = reps * ksteps * 28672
        print("%d WG/CU  %-9s row %4d B  %d stages: %6.1f B/clk per WG, %6.1f B/clk per CU  (%d cycles per step)" %
              (wgs_per_cu, "LDS-DMA" if mode == 0 else "registers", rb, stages, bytes_per_wg / c,
               wgs_per_cu * bytes_per_wg / c, c // (reps * ksteps)), flush=True)
