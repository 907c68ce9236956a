import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvalu_rate.so"))
lib.valu_rate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
out = torch.zeros(1024 * 1024, device="cuda")
cyc = torch.zeros(1, dtype=torch.int64, device="cuda")
names = ["v_dot2_f32_bf16 (1 useful MAC)", "v_fma_f32", "v_pk_fma_f32 (2 MAC)", "unpack + 2 v_fma (per bf16 pair)"]
iters = 2000
for threads, label in ((256, "1 wave/SIMD"), (512, "2 waves/SIMD"), (1024, "4 waves/SIMD")):
    for op in range(4):
        per_iter = {0: 64, 1: 64, 2: 32, 3: 32}[op]   # instruction groups per loop body
        lib.valu_rate(op, threads, 256, iters, out.data_ptr(), cyc.data_ptr())
        lib.valu_rate(op, threads, 256, iters, out.data_ptr(), cyc.data_ptr())
        c = int(cyc.item())
        print("%-14s %-34s %6.2f cycles per %s" % (label, names[op], c / (iters * per_iter),
              "instruction" if op < 3 else "pair (lshl+and+2 fma)"))
