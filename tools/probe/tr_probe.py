import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtr_probe.so"))
def run(addrs, label):
    a = torch.tensor(addrs, dtype=torch.int32, device="cuda")
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    lib.tr_probe_launch(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    o = out.cpu().view(64, 4).tolist()
    print(label)
    for l in (0, 1, 2, 3, 4, 5, 15, 16, 17, 31, 32, 48, 63):
        print("  lane %2d addr %5d (elem %4d) ->" % (l, addrs[l], addrs[l] // 2), o[l])
# A: lane-linear 8-byte pieces
run([l * 8 for l in range(64)], "A: addr = lane*8")
# B: within each 16-lane group a [4 rows][16 cols] block with row stride 256 B: lane i -> row i//4, cols (i%4)*4
run([(l // 16) * 2048 + ((l % 16) // 4) * 256 + ((l % 16) % 4) * 8 for l in range(64)], "B: 4x16 block, row stride 256 B, group stride 2048 B")
