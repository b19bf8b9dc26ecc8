"""oracle/ref_lib.py -- TEST INFRASTRUCTURE.  Python driver for oracle/_ref/libadcensus_ref.so
(the reference's own adcensus.cu compiled for gfx950, see ref_shim.hip / build_ref.py).

    ref = RefLib()                       # raises RefUnavailable if the .so was not built
    ref.call("cross", x0, out, L1, tau1) # == adcensus.cross(x0, out, L1, tau1) in main.lua
    d = ref.call("median2d", img, 5)[0]  # functions that push a new tensor return it

Arguments follow the reference's Lua call sites: torch CUDA float tensors are passed as
torch.CudaTensor, CPU tensors as torch.{Float,Double,Int,Long}Tensor, numbers as Lua numbers.
The reference launches on the NULL stream without synchronising; `call` synchronises the
device before and after so that it composes with work on torch's stream.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libadcensus_ref.so")

NIL, NUMBER, STRING, UDATA = 0, 1, 2, 3


class RefUnavailable(RuntimeError):
    pass


class RefLuaError(RuntimeError):
    """The reference raised luaL_error (checkCudaError or an argument check)."""


class _Val(C.Structure):
    _fields_ = [("tag", C.c_int), ("num", C.c_double), ("str", C.c_char_p), ("ud", C.c_void_p)]


_KIND = {"torch.float32": (1, b"torch.FloatTensor"), "torch.float64": (2, b"torch.DoubleTensor"),
         "torch.int32": (3, b"torch.IntTensor"), "torch.int64": (4, b"torch.LongTensor")}


class RefLib:
    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise RefUnavailable("%s not built (python oracle/build_ref.py, needs /root/reference)" % path)
        self.lib = lib = C.CDLL(path)
        lib.mcref_func_name.restype = C.c_char_p
        lib.mcref_tensor_new.restype = C.c_void_p
        lib.mcref_tensor_new.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_long)]
        lib.mcref_tensor_data.restype = C.c_void_p
        lib.mcref_tensor_data.argtypes = [C.c_void_p]
        lib.mcref_tensor_ndim.argtypes = [C.c_void_p]
        lib.mcref_tensor_size.restype = C.c_long
        lib.mcref_tensor_size.argtypes = [C.c_void_p, C.c_int]
        lib.mcref_tensor_free.argtypes = [C.c_void_p, C.c_int]
        lib.mcref_copy_d2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.mcref_call.argtypes = [C.c_char_p, C.POINTER(_Val), C.c_int, C.POINTER(_Val), C.POINTER(C.c_int),
                                   C.c_char_p, C.c_int]

    def functions(self):
        return [self.lib.mcref_func_name(i).decode() for i in range(self.lib.mcref_nfuncs())]

    def call(self, name, *args, table="adcensus"):
        import torch
        lib = self.lib
        vals = (_Val * max(1, len(args)))()
        handles = []
        keep = []
        for i, a in enumerate(args):
            if isinstance(a, torch.Tensor):
                if not a.is_contiguous():
                    raise TypeError("contiguous tensor expected")
                sizes = (C.c_long * max(1, a.dim()))(*a.shape)
                if a.is_cuda:
                    if a.dtype != torch.float32:
                        raise TypeError("torch.CudaTensor is float32")
                    kind, tname = 0, b"torch.CudaTensor"
                else:
                    kind, tname = _KIND[str(a.dtype)]
                h = lib.mcref_tensor_new(kind, C.c_void_p(a.data_ptr()), a.dim(), sizes)
                handles.append(h)
                keep.append((a, sizes, tname))
                vals[i] = _Val(UDATA, 0.0, tname, h)
            elif isinstance(a, str):
                b = a.encode()
                keep.append(b)
                vals[i] = _Val(STRING, 0.0, b, None)
            elif isinstance(a, (int, float)):
                vals[i] = _Val(NUMBER, float(a), None, None)
            else:
                raise TypeError("unsupported argument %r" % (a,))
        rets = (_Val * 8)()
        nret = C.c_int(0)
        err = C.create_string_buffer(512)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        rc = lib.mcref_call(("%s.%s" % (table, name)).encode(), vals, len(args), rets, C.byref(nret), err, 512)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        out = []
        try:
            if rc == 1:
                raise RefLuaError(err.value.decode())
            if rc != 0:
                raise RefUnavailable(err.value.decode())
            for i in range(nret.value):
                r = rets[i]
                if r.tag == NUMBER:
                    out.append(r.num)
                elif r.tag == UDATA:
                    tname = r.str.decode()
                    shape = [lib.mcref_tensor_size(r.ud, k) for k in range(lib.mcref_tensor_ndim(r.ud))]
                    if tname == "torch.CudaTensor":
                        t = torch.empty(shape, dtype=torch.float32, device="cuda")
                        if t.numel():
                            lib.mcref_copy_d2d(C.c_void_p(t.data_ptr()), C.c_void_p(lib.mcref_tensor_data(r.ud)),
                                               t.numel() * 4)
                        lib.mcref_tensor_free(r.ud, 1)
                    else:
                        dt = {"torch.FloatTensor": torch.float32, "torch.DoubleTensor": torch.float64,
                              "torch.IntTensor": torch.int32, "torch.LongTensor": torch.int64}[tname]
                        t = torch.empty(shape, dtype=dt)
                        if t.numel():
                            C.memmove(t.data_ptr(), lib.mcref_tensor_data(r.ud), t.numel() * t.element_size())
                        lib.mcref_tensor_free(r.ud, 0)
                    out.append(t)
        finally:
            for h in handles:
                lib.mcref_tensor_free(h, 0)
        return out
