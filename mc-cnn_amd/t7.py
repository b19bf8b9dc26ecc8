"""Torch7 serialisation (`torch.save(fname, obj, 'ascii')` / `torch.load`) -- the format of the reference's trained
nets (`main.lua:587-600` saves `{clean_net(net_te), clean_net(net_te2), opt}` / `{clean_net(net_te), opt}` in ASCII mode,
`main.lua:894-898` loads them).  Torch7 is a third-party dependency absent from /root/reference (no version pinned:
`README.md` just says "Install Torch"); this module restates the published on-disk layout of torch7's `File.lua`
(`readObject` / `writeObject`) and `generic/Tensor.c` / `generic/Storage.c` (`read` / `write`):

  object   := TYPE ...            TYPE: 0 nil | 1 number | 2 string | 3 table | 4 torch object | 5 boolean
                                        6 function | 7, 8 recursive function
  number   := double              string := int length, raw chars          boolean := int
  table    := int index, [int n, n x (key object, value object)]           (body only the first time an index is seen)
  torch    := int index, [string "V <n>", string class, payload]           (idem)
     Tensor payload  := int ndim, ndim x long size, ndim x long stride, long storageOffset (1-based), storage object
     Storage payload := long n, n x element
     any other class := one object (the table of its fields)
  ASCII mode: every read/write call leaves one '\\n' after its values, values of one call are separated by ' ';
  binary mode: little-endian int32 / int64 / float64, raw element arrays.

No real `.t7` file exists in the build image, so the reader is pinned only by round trips through the writer below and by
hand-written ASCII samples following the layout above (tests/test_main_host.py): parity with torch7 itself is UNPINNED.
"""
import io
import struct

import numpy as np

TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN, TYPE_FUNCTION, TYPE_LEGACY_RECUR, TYPE_RECUR = range(9)

_ELEM = {"Float": np.float32, "Double": np.float64, "Cuda": np.float32, "Long": np.int64, "Int": np.int32,
         "Short": np.int16, "Byte": np.uint8, "Char": np.int8, "CudaDouble": np.float64, "CudaLong": np.int64,
         "CudaInt": np.int32, "Half": np.float16, "CudaHalf": np.float16}


class T7Object:
    """A torch class instance that is not a tensor/storage (nn.Sequential, cudnn.SpatialConvolution, ...)."""

    def __init__(self, cls, fields):
        self.cls = cls
        self.fields = fields if isinstance(fields, dict) else {"_value": fields}

    def __getitem__(self, k):
        return self.fields[k]

    def get(self, k, default=None):
        return self.fields.get(k, default)

    def __repr__(self):
        return "T7Object(%s, %s)" % (self.cls, sorted(map(str, self.fields)))


class T7Function:
    def __init__(self, dumped, upvalues):
        self.dumped, self.upvalues = dumped, upvalues


def _kind(cls, suffix):
    """'torch.CudaTensor' -> 'Cuda' for suffix 'Tensor'; None if cls is not of that family."""
    if cls.startswith("torch.") and cls.endswith(suffix):
        k = cls[len("torch."):-len(suffix)]
        return k if k in _ELEM else None
    return None


class _Ascii:
    def __init__(self, data):
        self.b, self.p = data, 0

    def _eol(self):
        if self.p < len(self.b) and self.b[self.p:self.p + 1] == b"\n":
            self.p += 1

    def _tokens(self, n):
        out = []
        b, p = self.b, self.p
        for _ in range(n):
            while p < len(b) and b[p:p + 1].isspace():
                p += 1
            q = p
            while q < len(b) and not b[q:q + 1].isspace():
                q += 1
            if q == p:
                raise ValueError("t7: unexpected end of file")
            out.append(b[p:q])
            p = q
        self.p = p
        self._eol()
        return out

    def ints(self, n=1):
        return [int(t) for t in self._tokens(n)]

    longs = ints

    def doubles(self, n=1):
        return [float(t) for t in self._tokens(n)]

    def chars(self, n):
        s = self.b[self.p:self.p + n]
        if len(s) != n:
            raise ValueError("t7: unexpected end of file")
        self.p += n
        self._eol()
        return s

    def array(self, dtype, n):
        if n == 0:
            return np.zeros(0, dtype)
        if np.dtype(dtype).itemsize == 1:  # byte / char storages are raw in ASCII mode too
            return np.frombuffer(self.chars(n), dtype=dtype).copy()
        return np.array([float(t) for t in self._tokens(n)]).astype(dtype)


class _Binary:
    def __init__(self, data):
        self.b, self.p = data, 0

    def _take(self, n):
        s = self.b[self.p:self.p + n]
        if len(s) != n:
            raise ValueError("t7: unexpected end of file")
        self.p += n
        return s

    def ints(self, n=1):
        return list(struct.unpack("<%di" % n, self._take(4 * n)))

    def longs(self, n=1):
        return list(struct.unpack("<%dq" % n, self._take(8 * n)))

    def doubles(self, n=1):
        return list(struct.unpack("<%dd" % n, self._take(8 * n)))

    def chars(self, n):
        return self._take(n)

    def array(self, dtype, n):
        return np.frombuffer(self._take(n * np.dtype(dtype).itemsize), dtype=np.dtype(dtype).newbyteorder("<")).astype(dtype)


def _read(f, memo):
    t = f.ints()[0]
    if t == TYPE_NIL:
        return None
    if t == TYPE_NUMBER:
        v = f.doubles()[0]
        return int(v) if v == int(v) and abs(v) < 2 ** 53 else v
    if t == TYPE_BOOLEAN:
        return f.ints()[0] == 1
    if t == TYPE_STRING:
        return f.chars(f.ints()[0]).decode("latin-1")
    if t == TYPE_FUNCTION:
        dumped = f.chars(f.ints()[0])
        return T7Function(dumped, _read(f, memo))
    if t not in (TYPE_TABLE, TYPE_TORCH, TYPE_RECUR, TYPE_LEGACY_RECUR):
        raise ValueError("t7: unknown type id %d" % t)
    index = f.ints()[0]
    if index in memo:
        return memo[index]
    if t in (TYPE_RECUR, TYPE_LEGACY_RECUR):
        fn = T7Function(f.chars(f.ints()[0]), None)
        memo[index] = fn
        fn.upvalues = _read(f, memo)
        return fn
    if t == TYPE_TABLE:
        tab = {}
        memo[index] = tab
        for _ in range(f.ints()[0]):
            k = _read(f, memo)
            tab[k] = _read(f, memo)
        return tab
    version = f.chars(f.ints()[0]).decode("latin-1")
    cls = f.chars(f.ints()[0]).decode("latin-1") if version.startswith("V ") else version
    kind = _kind(cls, "Tensor")
    if kind is not None:
        ndim = f.ints()[0]
        size, stride = f.longs(ndim), f.longs(ndim)
        offset = f.longs()[0] - 1
        holder = [None]
        memo[index] = holder  # (a tensor cannot contain itself; the slot keeps the index taken)
        storage = _read(f, memo)
        if storage is None or ndim == 0:
            arr = np.zeros([0], _ELEM[kind])
        else:
            arr = np.lib.stride_tricks.as_strided(storage[offset:], shape=size, strides=[s * storage.itemsize for s in stride]).copy()
        memo[index] = arr
        return arr
    kind = _kind(cls, "Storage")
    if kind is not None:
        arr = f.array(_ELEM[kind], f.longs()[0])
        memo[index] = arr
        return arr
    obj = T7Object(cls, {})
    memo[index] = obj
    fields = _read(f, memo)
    obj.fields = fields if isinstance(fields, dict) else {"_value": fields}
    return obj


def load(path_or_bytes, mode=None):
    """torch.load(fname, mode): mode 'ascii' | 'binary' | None (guess: an ASCII file starts with a digit and a newline)."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    if mode is None:
        mode = "ascii" if data[:1].isdigit() and data[1:2] in (b"\n", b" ") else "binary"
    return _read(_Ascii(bytes(data)) if mode == "ascii" else _Binary(bytes(data)), {})


# ---- writer (ASCII): what torch.save(fname, obj, 'ascii') emits for numbers, strings, booleans, tables, tensors and
# ---- T7Object instances; used to build test files and to export seeded nets for a Torch7 host -----------------------------
class _Writer:
    def __init__(self):
        self.out = io.BytesIO()
        self.index = {}
        self.next = 1

    def line(self, *vals):
        self.out.write((" ".join(str(v) for v in vals) + "\n").encode("latin-1"))

    def string(self, s):
        b = s.encode("latin-1")
        self.line(len(b))
        self.out.write(b + b"\n")

    def number_repr(self, v):
        return repr(float(v)) if float(v) != int(v) else str(int(v))

    def obj(self, o):
        if o is None:
            self.line(TYPE_NIL)
        elif isinstance(o, bool):
            self.line(TYPE_BOOLEAN)
            self.line(1 if o else 0)
        elif isinstance(o, (int, float, np.integer, np.floating)):
            self.line(TYPE_NUMBER)
            self.line(self.number_repr(o))
        elif isinstance(o, str):
            self.line(TYPE_STRING)
            self.string(o)
        elif isinstance(o, (dict, list, tuple)):
            self.line(TYPE_TABLE)
            if self._seen(o):
                return
            items = list(o.items()) if isinstance(o, dict) else [(i + 1, v) for i, v in enumerate(o)]
            self.line(len(items))
            for k, v in items:
                self.obj(k)
                self.obj(v)
        elif isinstance(o, np.ndarray):
            kind = {np.dtype(np.float32): "Float", np.dtype(np.float64): "Double", np.dtype(np.int64): "Long",
                    np.dtype(np.int32): "Int", np.dtype(np.uint8): "Byte"}[o.dtype]
            self.line(TYPE_TORCH)
            if self._seen(o):
                return
            self.string("V 1")
            self.string("torch.%sTensor" % kind)
            a = np.ascontiguousarray(o)
            self.line(a.ndim)
            if a.ndim:
                self.line(*a.shape)
                self.line(*[s // a.itemsize for s in a.strides])
            self.line(1)
            if a.size == 0:
                self.line(TYPE_NIL)
                return
            self.line(TYPE_TORCH)
            self.line(self._take())
            self.string("V 1")
            self.string("torch.%sStorage" % kind)
            self.line(a.size)
            if a.dtype == np.uint8:
                self.out.write(a.tobytes() + b"\n")
            else:
                self.line(*[repr(float(x)) if a.dtype.kind == "f" else int(x) for x in a.ravel()])
        elif isinstance(o, T7Object):
            self.line(TYPE_TORCH)
            if self._seen(o):
                return
            self.string("V 1")
            self.string(o.cls)
            self.obj(o.fields)
        else:
            raise TypeError("t7: cannot serialise %r" % type(o))

    def _take(self):
        i = self.next
        self.next += 1
        return i

    def _seen(self, o):
        if id(o) in self.index:
            self.line(self.index[id(o)])
            return True
        self.index[id(o)] = self._take()
        self.line(self.index[id(o)])
        return False


def dumps(obj):
    w = _Writer()
    w.obj(obj)
    return w.out.getvalue()


def save(path, obj):
    open(path, "wb").write(dumps(obj))


# ---- nets of the reference ------------------------------------------------------------------------------------------------
def _modules(seq):
    mods = seq["modules"]
    return [mods[i] for i in sorted(k for k in mods if isinstance(k, int))]


def conv_layers(net_te):
    """[(w (out,in,kh,kw), b (out))] of the (cudnn|nn).SpatialConvolution modules of net_te, in order."""
    out = []
    for m in _modules(net_te):
        if isinstance(m, T7Object) and m.cls.endswith("SpatialConvolution"):
            w = np.asarray(m["weight"], np.float32)
            b = np.asarray(m["bias"], np.float32).reshape(-1)
            nout = int(m.get("nOutputPlane", w.shape[0]))
            kh, kw = int(m.get("kH", 3)), int(m.get("kW", 3))
            out.append((w.reshape(nout, -1, kh, kw), b))
    return out


def fc_layers(net_te2):
    """[(w (out,in), b (out))] of the nn.SpatialConvolution1_fw modules of net_te2 (SpatialConvolution1_fw.lua:7-8)."""
    out = []
    for m in _modules(net_te2):
        if isinstance(m, T7Object) and m.cls == "nn.SpatialConvolution1_fw":
            w = np.asarray(m["weight"], np.float32)
            out.append((w, np.asarray(m["bias"], np.float32).reshape(-1)))
    return out


def load_reference_net(path, arch):
    """main.lua:894-905: arch slow -> (conv layers of obj[1], fc layers of obj[2]); arch fast -> (conv layers of obj[1]
    -- its trailing Normalize2 / StereoJoin1 modules carry no weights -- , None)."""
    obj = load(path)
    net_te = obj[1]
    if arch == "slow":
        return conv_layers(net_te), fc_layers(obj[2])
    return conv_layers(net_te), None
