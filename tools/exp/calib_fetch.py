import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcalib_fetch.so"))
buf = torch.randn(1 << 28, device="cuda")  # 1 GiB
out = torch.zeros(4, device="cuda")
torch.cuda.synchronize()
lib.calib(ctypes.c_void_p(buf.data_ptr()), ctypes.c_size_t(buf.numel() * 4), ctypes.c_void_p(out.data_ptr()))
