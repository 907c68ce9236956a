"""Diagnostic: the SAME operand bytes, each read once per pass, row-major (224 rows x 128 B per K-step, 3072 B apart)
against a blocked layout ([panel][K-step][224 rows][64 elements]: one 28 KiB tile per K-step is contiguous)."""
import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfill_rate.so"))
lib.fill_rate2.argtypes = [C.c_int] * 3 + [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
rows, K, ksteps, reps = 224 * 58, 1536, 24, 20
A = torch.randn(rows, K, device="cuda").bfloat16()
blocked = A.view(58, 224, 24, 64).permute(0, 2, 1, 3).contiguous()       # [panel][kstep][row][64]
cyc = torch.zeros(1, dtype=torch.int64, device="cuda")
sink = torch.zeros(4, device="cuda")
for wgs_per_cu in (1, 2):
    for stages in (2, 3, 4):
        for name, mode, args in (("row-major, LDS-DMA", 0, (128, stages, A.data_ptr(), K, rows, ksteps, reps, 256 * wgs_per_cu, 0, 0)),
                                 ("blocked,   LDS-DMA", 0, (1024, stages, blocked.data_ptr(), 512, 58 * 24 * 28, ksteps, reps, 256 * wgs_per_cu, 14336, 672)),
                                 ("row-major, registers", 1, (128, 2, A.data_ptr(), K, rows, ksteps, reps, 256 * wgs_per_cu, 0, 0)),
                                 ("blocked,   registers", 1, (1024, 2, blocked.data_ptr(), 512, 58 * 24 * 28, ksteps, reps, 256 * wgs_per_cu, 14336, 672))):
            if mode == 1 and stages != 2:
                continue
            if stages == 4 and args[0] not in (128, 1024):
                continue
            rb, st, ptr, ld, nrows, ks, rp, wgs, kse, pr = args
            for _ in range(2):
                rc = lib.fill_rate2(mode, rb, st, ptr, ld, nrows, ks, rp, wgs, cyc.data_ptr(), sink.data_ptr(), kse, pr)
                torch.cuda.synchronize()
            assert rc == 0, rc
            c = int(cyc.item())
            b = reps * ksteps * 28672
            res = "co-resident" if st * 28672 * wgs_per_cu <= 160 * 1024 else "NOT co-resident (LDS)"
            print("%d WG/CU %-22s %d stages: %6.1f B/clk per WG   %s" % (wgs_per_cu, name, st, b / c, res), flush=True)
