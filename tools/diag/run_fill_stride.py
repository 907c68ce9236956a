"""Diagnostic: LDS fill rate of 128-B tile rows as a function of the row stride (same kernel as run_fill_rate.py)."""
import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfill_rate.so"))
lib.fill_rate.argtypes = [C.c_int] * 3 + [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
buf = torch.randn(48 * 1024 * 1024, device="cuda").bfloat16()        # 96 MB
cyc = torch.zeros(1, dtype=torch.int64, device="cuda")
sink = torch.zeros(4, device="cuda")
reps, ksteps = 20, 24
for wgs_per_cu in (1, 2):
    for ld in (64, 128, 256, 512, 1024, 1536, 1600, 2048, 4096, 8192):
        # a K-step advances by 64 elements along the row when the row is long enough, else by whole tiles
        rows = 224 * 58
        if ld < 64 * ksteps:
            continue_ok = False
        need = rows * ld + 64 * ksteps
        if need > buf.numel():
            rows = (buf.numel() - 64 * ksteps) // ld // 224 * 224
        for _ in range(2):
            rc = lib.fill_rate(0, 128, 2, buf.data_ptr(), ld, rows, ksteps, reps, 256 * wgs_per_cu, cyc.data_ptr(), sink.data_ptr())
            torch.cuda.synchronize()
        assert rc == 0
        c = int(cyc.item())
        b = reps * ksteps * 28672
        print("%d WG/CU  row stride %6d B (%5d rows in play): %6.1f B/clk per WG  %6.1f per CU" %
              (wgs_per_cu, ld * 2, rows, b / c, wgs_per_cu * b / c), flush=True)
