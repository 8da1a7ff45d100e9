"""A small SPIR-V interpreter: test infrastructure that executes the reference's COMMITTED compute-shader binaries
(`/root/reference/shaders/**/*.comp.spv`) one invocation at a time, literally, so that the CPU oracle can be pinned against
the dataflow of the reference's own shaders (tests/golden/make_spirv_vectors.py writes the fixtures, tests/test_spirv_pin.py
checks the oracle against them). The .spv files are READ AS DATA: nothing from them runs natively, nothing of them is copied
into this repository; the interpreter only runs where the reference checkout exists (the build container).

What "literally" means
  * every OpFAdd / OpFSub / OpFMul / OpFDiv / OpFNegate is one IEEE binary32 operation (numpy.float32), evaluated in program
    order, never fused: SPIR-V permits contraction (no NoContraction decoration in these modules), a literal execution is
    the uncontracted member of the permitted class. The oracle is compared with its contraction rule switched off
    (oracle/libszg_oracle_literal.so, -DSZG_ORACLE_LITERAL).
  * OpDot, OpMatrixTimesVector, OpMatrixTimesMatrix, OpVectorTimesScalar: products summed left to right, component 0 first.
  * GLSL.std.450: Sqrt correctly rounded; FAbs / FMin / FMax / FClamp exact (clamp = min(max(x, lo), hi)); Length = sqrt(dot),
    Distance = length(a - b), Normalize = v * (1 / sqrt(dot(v, v))), FMix = x * (1 - a) + y * a, SmoothStep by its
    definition, all in binary32 (the conventions the oracle states). Exp / Pow / Sin / Cos / Asin / Acos are
    implementation-defined within a ULP budget in Vulkan: the `builtins` object supplies them (the pinned algorithms of
    include/szg/fpmath.h through oracle.binding.builtin_eval, so that both sides use the same ones).
  * images: `Image` objects supplied by the caller implement fetch / store / sample; the sampler models (LINEAR with
    clamp-to-edge in binary32, NEAREST, border colours) are the caller's, stated in make_spirv_vectors.py.
  * PhysicalStorageBuffer pointers are integer addresses into `Memory`, decoded with the module's own Offset / ArrayStride /
    MatrixStride decorations.

Supported: exactly the instruction set of the four hot-path shaders and the two OETF shaders and the three raster-pass shaders (86 opcodes, 18 extended instructions); anything else
raises NotImplementedError naming the opcode.
"""
import struct

import numpy as np

F32 = np.float32


def clone(v):
    """Copy of a composite value (nested lists of scalars); scalars, images and pointers are shared."""
    return [clone(x) for x in v] if isinstance(v, list) else v

_ERR = dict(over="ignore", invalid="ignore", divide="ignore", under="ignore")

# storage classes
SC_UNIFORM_CONSTANT, SC_INPUT, SC_OUTPUT, SC_PRIVATE, SC_FUNCTION, SC_PUSH_CONSTANT, SC_PSB = 0, 1, 3, 6, 7, 9, 5349
# decorations
DEC_BUILTIN, DEC_BINDING, DEC_SET, DEC_OFFSET, DEC_ARRAY_STRIDE, DEC_MATRIX_STRIDE, DEC_COL_MAJOR, DEC_ROW_MAJOR = 11, 33, 34, 35, 6, 7, 5, 4
DEC_NO_CONTRACTION = 42
BUILTIN_GLOBAL_INVOCATION_ID, BUILTIN_VERTEX_INDEX, BUILTIN_INSTANCE_INDEX = 28, 42, 43


class Type:
    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)

    def __repr__(self):
        return f"Type({self.kind}, {{k: v for k, v in self.__dict__.items() if k != 'kind'}})"


class Ref:
    """Pointer into interpreter-owned storage: a root list cell and a path of indices."""

    __slots__ = ("root", "path", "name")

    def __init__(self, root, path=(), name=None):
        self.root, self.path, self.name = root, path, name

    def load(self):
        v = self.root[0]
        for i in self.path:
            v = v[i]
        return v

    def store(self, value):
        if not self.path:
            self.root[0] = value
            return
        v = self.root[0]
        for i in self.path[:-1]:
            v = v[i]
        v[self.path[-1]] = value


class PsbPtr:
    """PhysicalStorageBuffer pointer: address + pointee type id."""

    __slots__ = ("address", "type_id")

    def __init__(self, address, type_id):
        self.address, self.type_id = address, type_id


class Memory:
    """Fake device memory: named allocations at fixed addresses."""

    def __init__(self):
        self.blocks = []  # (base, bytes)
        self.next = 0x10000

    def alloc(self, data: bytes):
        base = self.next
        self.blocks.append((base, bytes(data)))
        self.next = (base + len(data) + 0xFFF) & ~0xFFF
        self.next += 0x1000
        return base

    def read(self, address, size):
        for base, data in self.blocks:
            if base <= address and address + size <= base + len(data):
                return data[address - base: address - base + size]
        raise RuntimeError(f"out-of-bounds device read of {size} bytes at {address:#x}")


class Function:
    def __init__(self, fid, type_id, result_type):
        self.id, self.type_id, self.result_type = fid, type_id, result_type
        self.params = []
        self.blocks = {}  # label -> list of (op, operands)
        self.entry = None


class Module:
    def __init__(self, path):
        raw = open(path, "rb").read()
        w = struct.unpack("<%dI" % (len(raw) // 4), raw)
        assert w[0] == 0x07230203, "not a SPIR-V module"
        self.bound = w[3]
        self.types, self.consts, self.names, self.member_names = {}, {}, {}, {}
        self.decor, self.member_decor = {}, {}
        self.globals = {}  # id -> (pointer type id, storage class)
        self.functions = {}
        self.entry_point = None
        self.local_size = None
        self.ext_imports = {}
        self.no_contraction = 0
        i = 5
        fn = block = None
        while i < len(w):
            wc, op = w[i] >> 16, w[i] & 0xFFFF
            a = w[i + 1: i + wc]
            i += wc
            if op in (3, 4, 10, 17, 14):  # Source, SourceExtension, Extension, Capability, MemoryModel
                continue
            if op == 5:
                self.names[a[0]] = self._string(a[1:])
            elif op == 6:
                self.member_names[(a[0], a[1])] = self._string(a[2:])
            elif op == 11:
                self.ext_imports[a[0]] = self._string(a[1:])
            elif op == 15:
                self.entry_point = a[1]
            elif op == 16:
                if a[1] == 17:  # LocalSize
                    self.local_size = tuple(a[2:5])
            elif op == 71:
                self.decor.setdefault(a[0], {})[a[1]] = a[2:]
                if a[1] == DEC_NO_CONTRACTION:
                    self.no_contraction += 1
            elif op == 72:
                self.member_decor.setdefault((a[0], a[1]), {})[a[2]] = a[3:]
            elif op == 19:
                self.types[a[0]] = Type("void")
            elif op == 20:
                self.types[a[0]] = Type("bool")
            elif op == 21:
                self.types[a[0]] = Type("int", width=a[1], signed=bool(a[2]))
            elif op == 22:
                assert a[1] == 32, "only 32-bit floats occur in these shaders"
                self.types[a[0]] = Type("float", width=a[1])
            elif op == 23:
                self.types[a[0]] = Type("vector", elem=a[1], n=a[2])
            elif op == 24:
                self.types[a[0]] = Type("matrix", col=a[1], n=a[2])
            elif op == 25:
                self.types[a[0]] = Type("image", sampled_type=a[1], dim=a[2], depth=a[3], arrayed=a[4], ms=a[5], sampled=a[6], format=a[7])
            elif op == 26:
                self.types[a[0]] = Type("sampler")
            elif op == 27:
                self.types[a[0]] = Type("sampled_image", image=a[1])
            elif op == 28:
                self.types[a[0]] = Type("array", elem=a[1], length_id=a[2])
            elif op == 29:
                self.types[a[0]] = Type("runtime_array", elem=a[1])
            elif op == 30:
                self.types[a[0]] = Type("struct", members=list(a[1:]))
            elif op == 32:
                self.types[a[0]] = Type("pointer", storage=a[1], pointee=a[2])
            elif op == 33:
                self.types[a[0]] = Type("function", ret=a[1], params=list(a[2:]))
            elif op == 39:
                pass  # forward pointer: the OpTypePointer follows
            elif op == 41:
                self.consts[a[1]] = True
            elif op == 42:
                self.consts[a[1]] = False
            elif op == 43:
                t = self.types[a[0]]
                if t.kind == "float":
                    self.consts[a[1]] = F32(struct.unpack("<f", struct.pack("<I", a[2]))[0])
                elif t.width == 64:
                    self.consts[a[1]] = a[2] | (a[3] << 32)
                else:
                    self.consts[a[1]] = a[2]
            elif op == 44:
                self.consts[a[1]] = [self.consts[c] for c in a[2:]]
            elif op == 59 and fn is None:
                self.globals[a[1]] = (a[0], a[2])
            elif op == 54:
                fn = Function(a[1], a[3], a[0])
                self.functions[a[1]] = fn
            elif op == 55:
                fn.params.append(a[1])
            elif op == 56:
                fn = block = None
            elif op == 248:
                block = []
                fn.blocks[a[0]] = block
                if fn.entry is None:
                    fn.entry = a[0]
            elif fn is not None:
                block.append((op, a))
            else:
                raise NotImplementedError(f"module-level opcode {op}")

    @staticmethod
    def _string(words):
        b = b"".join(struct.pack("<I", x) for x in words)
        return b.split(b"\0")[0].decode()

    # ---- explicit layout (PhysicalStorageBuffer / PushConstant) ----------------------------------------------------
    def decode(self, mem, address, type_id, layout=None):
        t = self.types[type_id]
        k = t.kind
        if k == "float":
            return F32(struct.unpack("<f", mem.read(address, 4))[0])
        if k == "int":
            if t.width == 64:
                return struct.unpack("<Q", mem.read(address, 8))[0]
            return struct.unpack("<I", mem.read(address, 4))[0]
        if k == "vector":
            size = 4 if self.types[t.elem].kind != "int" or self.types[t.elem].width == 32 else 8
            return [self.decode(mem, address + size * c, t.elem) for c in range(t.n)]
        if k == "matrix":
            stride = layout[DEC_MATRIX_STRIDE][0]
            assert DEC_COL_MAJOR in layout, "row-major matrices do not occur in these shaders"
            return [self.decode(mem, address + stride * c, t.col) for c in range(t.n)]
        if k == "struct":
            return [self.decode(mem, address + self.member_decor[(type_id, m)][DEC_OFFSET][0], mt, self.member_decor[(type_id, m)])
                    for m, mt in enumerate(t.members)]
        if k == "array":
            stride = self.decor[type_id][DEC_ARRAY_STRIDE][0]
            return [self.decode(mem, address + stride * e, t.elem, layout) for e in range(self.consts[t.length_id])]
        if k == "pointer":
            return PsbPtr(struct.unpack("<Q", mem.read(address, 8))[0], t.pointee)
        raise NotImplementedError(f"decode of {k}")


class Image:
    """Base class of what the caller binds to a descriptor: override what the shader uses."""

    def size(self):
        raise NotImplementedError

    def fetch(self, x, y):  # OpImageRead
        raise NotImplementedError

    def store(self, x, y, texel):  # OpImageWrite
        raise NotImplementedError

    def sample(self, sampler, u, v):  # OpImageSampleExplicitLod, Lod 0
        raise NotImplementedError


class SampledImage:
    def __init__(self, image, sampler):
        self.image, self.sampler = image, sampler


class Interpreter:
    def __init__(self, module, memory, builtins, push_constants, descriptors):
        """push_constants: bytes of the push-constant block; descriptors: {(set, binding): object or list of objects}."""
        self.m, self.mem, self.b = module, memory, builtins
        self.pc_mem = Memory()
        self.pc_base = self.pc_mem.alloc(push_constants + b"\0" * 16)
        self.descriptors = descriptors
        self.executed = 0
        self.trace = None  # a list: (function name, variable name, value) of every OpStore to a named variable

    # ---- helpers ----------------------------------------------------------------------------------------------------
    def _zero(self, type_id):
        t = self.m.types[type_id]
        k = t.kind
        if k == "float":
            return F32(0)
        if k == "int":
            return 0
        if k == "bool":
            return False
        if k == "vector":
            return [self._zero(t.elem) for _ in range(t.n)]
        if k == "matrix":
            return [self._zero(t.col) for _ in range(t.n)]
        if k == "struct":
            return [self._zero(mt) for mt in t.members]
        if k == "array":
            return [self._zero(t.elem) for _ in range(self.m.consts[t.length_id])]
        if k == "pointer":
            return None
        raise NotImplementedError(k)

    def _mask(self, v, type_id):
        t = self.m.types[type_id]
        if t.kind == "vector":
            t = self.m.types[t.elem]
        return v & ((1 << t.width) - 1)

    @staticmethod
    def _map(f, *xs):
        if isinstance(xs[0], list):
            return [f(*c) for c in zip(*xs)]
        return f(*xs)

    def _dot(self, a, b):
        acc = a[0] * b[0]
        for x, y in zip(a[1:], b[1:]):
            acc = acc + x * y
        return acc

    def run(self, global_id=None, inputs=None, derivatives=None):
        """One invocation. Compute: `global_id`. Vertex / fragment stages: `inputs` maps a variable NAME (or a BuiltIn
        number) to its value; the Output variables are left in `self.outputs` by name (gl_PerVertex members as a list).
        `derivatives`: the values OpDPdx / OpDPdy return, in execution order (fixed function, supplied by the caller)."""
        m = self.m
        self.globals = {}
        self.outputs = {}
        self.derivatives = list(derivatives or [])
        inputs = inputs or {}
        for gid, (ptype, storage) in m.globals.items():
            pointee = m.types[ptype].pointee
            d = m.decor.get(gid, {})
            if storage == SC_INPUT:
                builtin = d.get(DEC_BUILTIN, [None])[0]
                if builtin == BUILTIN_GLOBAL_INVOCATION_ID:
                    self.globals[gid] = Ref([list(global_id)])
                elif builtin is not None:
                    self.globals[gid] = Ref([inputs[builtin]])
                else:
                    self.globals[gid] = Ref([clone(inputs[m.names[gid]])])
            elif storage == SC_OUTPUT:
                cell = [self._zero(pointee)]
                self.globals[gid] = Ref(cell)
                self.outputs[m.names.get(gid) or "gl_PerVertex"] = cell
            elif storage == SC_PUSH_CONSTANT:
                self.globals[gid] = ("pc", self.pc_base, pointee)
            elif storage == SC_UNIFORM_CONSTANT:
                self.globals[gid] = Ref([self.descriptors[(d[DEC_SET][0], d[DEC_BINDING][0])]])
            elif storage == SC_PRIVATE:
                self.globals[gid] = Ref([self._zero(pointee)])
            else:
                raise NotImplementedError(f"global in storage class {storage}")
        with np.errstate(**_ERR):
            self.call(m.functions[m.entry_point], [])

    # ---- execution --------------------------------------------------------------------------------------------------
    def call(self, fn, args):
        m = self.m
        T = m.types
        C = m.consts
        V = dict(zip(fn.params, args))
        G = self.globals

        def val(i):
            if i in V:
                return V[i]
            if i in C:
                return C[i]
            if i in G:
                return G[i]
            raise KeyError(f"id %{i} ({m.names.get(i)}) has no value")

        label, prev = fn.entry, None
        while True:
            block = fn.blocks[label]
            # OpPhi first (all read their operands from the predecessor's state)
            nxt = None
            for op, a in block:
                self.executed += 1
                if op == 245:  # Phi
                    pairs = a[2:]
                    for k in range(0, len(pairs), 2):
                        if pairs[k + 1] == prev:
                            V[a[1]] = val(pairs[k])
                            break
                    else:
                        raise RuntimeError("OpPhi without a matching predecessor")
                elif op == 59:  # Variable (Function)
                    init = clone(val(a[3])) if len(a) > 3 else self._zero(T[a[0]].pointee)
                    V[a[1]] = Ref([init], (), m.names.get(a[1]))
                elif op == 61:  # Load
                    p = val(a[2])
                    V[a[1]] = self._load(p, a[0])
                elif op == 62:  # Store
                    p = val(a[0])
                    v = val(a[1])
                    p.store(clone(v))
                    if self.trace is not None and isinstance(p, Ref) and p.name:
                        self.trace.append((m.names.get(fn.id), p.name + "".join(f"[{i}]" for i in p.path), clone(v)))
                elif op == 65:  # AccessChain
                    V[a[1]] = self._access_chain(val(a[2]), [val(x) for x in a[3:]], a[0])
                elif op == 57:  # FunctionCall
                    V[a[1]] = self.call(m.functions[a[2]], [val(x) for x in a[3:]])
                elif op == 12:  # ExtInst
                    V[a[1]] = self._ext(a[3], [val(x) for x in a[4:]])
                elif op == 79:  # VectorShuffle
                    both = list(val(a[2])) + list(val(a[3]))
                    V[a[1]] = [both[k] for k in a[4:]]
                elif op == 80:  # CompositeConstruct
                    t = T[a[0]]
                    parts = [val(x) for x in a[2:]]
                    if t.kind == "vector":
                        flat = []
                        for p in parts:
                            flat.extend(p if isinstance(p, list) else [p])
                        V[a[1]] = flat
                    else:
                        V[a[1]] = [clone(p) for p in parts]
                elif op == 81:  # CompositeExtract
                    v = val(a[2])
                    for k in a[3:]:
                        v = v[k]
                    V[a[1]] = v
                elif op == 127:
                    V[a[1]] = self._map(lambda x: -x, val(a[2]))
                elif op == 129:
                    V[a[1]] = self._map(lambda x, y: x + y, val(a[2]), val(a[3]))
                elif op == 131:
                    V[a[1]] = self._map(lambda x, y: x - y, val(a[2]), val(a[3]))
                elif op == 133:
                    V[a[1]] = self._map(lambda x, y: x * y, val(a[2]), val(a[3]))
                elif op == 136:
                    V[a[1]] = self._map(lambda x, y: x / y, val(a[2]), val(a[3]))
                elif op == 142:  # VectorTimesScalar
                    s = val(a[3])
                    V[a[1]] = [x * s for x in val(a[2])]
                elif op == 145:  # MatrixTimesVector: sum over columns of column * v[c], left to right
                    M, v = val(a[2]), val(a[3])
                    rows = len(M[0])
                    out = []
                    for r in range(rows):
                        acc = M[0][r] * v[0]
                        for c in range(1, len(M)):
                            acc = acc + M[c][r] * v[c]
                        out.append(acc)
                    V[a[1]] = out
                elif op == 146:  # MatrixTimesMatrix: column c of the result = A * (column c of B)
                    A, B = val(a[2]), val(a[3])
                    res = []
                    for bc in B:
                        col = []
                        for r in range(len(A[0])):
                            acc = A[0][r] * bc[0]
                            for k in range(1, len(A)):
                                acc = acc + A[k][r] * bc[k]
                            col.append(acc)
                        res.append(col)
                    V[a[1]] = res
                elif op == 148:
                    V[a[1]] = self._dot(val(a[2]), val(a[3]))
                elif op == 128:  # IAdd
                    V[a[1]] = self._map(lambda x, y: self._mask(x + y, a[0]), val(a[2]), val(a[3]))
                elif op == 110:  # ConvertFToS (round toward zero; out-of-range is undefined in SPIR-V and must not happen)
                    def ftos(x):
                        assert np.isfinite(x) and abs(float(x)) < 2 ** 31, "ConvertFToS of an out-of-range value"
                        return self._mask(int(x), a[0])
                    V[a[1]] = self._map(ftos, val(a[2]))
                elif op == 111:  # ConvertSToF
                    V[a[1]] = self._map(lambda x: F32(self._signed_scalar(x)), val(a[2]))
                elif op == 112:  # ConvertUToF
                    V[a[1]] = self._map(lambda x: F32(x), val(a[2]))
                elif op == 124:  # Bitcast (uint64 <-> pointer, float <-> int)
                    V[a[1]] = self._bitcast(val(a[2]), a[0])
                elif op == 167:
                    V[a[1]] = self._map(lambda x, y: x and y, val(a[2]), val(a[3]))
                elif op == 168:
                    V[a[1]] = self._map(lambda x: not x, val(a[2]))
                elif op == 169:  # Select
                    c, x, y = val(a[2]), val(a[3]), val(a[4])
                    V[a[1]] = [xx if cc else yy for cc, xx, yy in zip(c, x, y)] if isinstance(c, list) else (x if c else y)
                elif op == 176:
                    V[a[1]] = self._map(lambda x, y: x < y, val(a[2]), val(a[3]))
                elif op in (177, 179):
                    sg = self._signed_scalar
                    f = (lambda x, y: sg(x) < sg(y)) if op == 177 else (lambda x, y: sg(x) <= sg(y))
                    V[a[1]] = self._map(f, val(a[2]), val(a[3]))
                elif op == 180:
                    V[a[1]] = self._map(lambda x, y: bool(x == y), val(a[2]), val(a[3]))
                elif op == 184:
                    V[a[1]] = self._map(lambda x, y: bool(x < y), val(a[2]), val(a[3]))
                elif op == 188:
                    V[a[1]] = self._map(lambda x, y: bool(x <= y), val(a[2]), val(a[3]))
                elif op == 186:
                    V[a[1]] = self._map(lambda x, y: bool(x > y), val(a[2]), val(a[3]))
                elif op == 190:
                    V[a[1]] = self._map(lambda x, y: bool(x >= y), val(a[2]), val(a[3]))
                elif op == 86:  # SampledImage
                    V[a[1]] = SampledImage(val(a[2]), val(a[3]))
                elif op == 100:  # Image
                    s = val(a[2])
                    V[a[1]] = s.image if isinstance(s, SampledImage) else s
                elif op in (87, 88):  # ImageSampleImplicitLod (single-level images: the level is 0) / ImageSampleExplicitLod
                    s = val(a[2])
                    coord = val(a[3])
                    assert op == 87 or (a[4] == 2 and float(val(a[5])) == 0.0), "only Lod 0 occurs"
                    img, smp = (s.image, s.sampler) if isinstance(s, SampledImage) else (s, None)
                    V[a[1]] = [F32(x) for x in img.sample(smp, coord[0], coord[1])]
                elif op == 98:  # ImageRead
                    coord = val(a[3])
                    V[a[1]] = [F32(x) for x in val(a[2]).fetch(self._signed_scalar(coord[0]), self._signed_scalar(coord[1]))]
                elif op == 99:  # ImageWrite
                    coord = val(a[1])
                    val(a[0]).store(self._signed_scalar(coord[0]), self._signed_scalar(coord[1]), list(val(a[2])))
                elif op in (103, 104):  # ImageQuerySizeLod / ImageQuerySize
                    s = val(a[2])
                    img = s.image if isinstance(s, SampledImage) else s
                    V[a[1]] = list(img.size())
                elif op in (207, 208):  # DPdx / DPdy: fixed function, the caller's values in execution order
                    V[a[1]] = clone(self.derivatives.pop(0))
                elif op in (246, 247):  # LoopMerge / SelectionMerge
                    pass
                elif op == 249:
                    nxt = a[0]
                elif op == 250:
                    nxt = a[1] if val(a[0]) else a[2]
                elif op == 253:
                    return None
                elif op == 254:
                    return clone(val(a[0]))
                elif op == 255:
                    raise RuntimeError("OpUnreachable executed")
                else:
                    raise NotImplementedError(f"opcode {op}")
            assert nxt is not None, "block without terminator"
            prev, label = label, nxt

    # ---- pieces -----------------------------------------------------------------------------------------------------
    @staticmethod
    def _signed_scalar(x):
        return x - (1 << 32) if x >> 31 else x

    def _bitcast(self, v, result_type):
        t = self.m.types[result_type]
        if t.kind == "pointer":
            return PsbPtr(int(v), t.pointee)
        if isinstance(v, PsbPtr):
            return v.address
        if t.kind == "float":
            return F32(struct.unpack("<f", struct.pack("<I", v))[0])
        if t.kind == "int" and isinstance(v, np.floating):
            return struct.unpack("<I", struct.pack("<f", v))[0]
        if t.kind == "vector":
            return [self._bitcast(x, t.elem) for x in v]
        return v

    def _load(self, p, result_type):
        if isinstance(p, Ref):
            return clone(p.load())
        if isinstance(p, PsbPtr):
            return self.m.decode(self.mem, p.address, result_type, getattr(p, "layout", None))
        if isinstance(p, tuple) and p[0] in ("pc", "psb"):
            mem = self.pc_mem if p[0] == "pc" else self.mem
            return self.m.decode(mem, p[1], result_type, p[3] if len(p) > 3 else None)
        raise NotImplementedError(f"load through {p!r}")

    def _access_chain(self, base, indices, result_ptr_type):
        m = self.m
        if isinstance(base, Ref):
            return Ref(base.root, base.path + tuple(int(i) for i in indices), base.name)
        # explicit-layout pointers: ("pc" | "psb", address, pointee type id[, layout of a matrix member])
        if isinstance(base, PsbPtr):
            kind, address, type_id, layout = "psb", base.address, base.type_id, None
        else:
            kind, address, type_id = base[0], base[1], base[2]
            layout = base[3] if len(base) > 3 else None
        for idx in indices:
            t = m.types[type_id]
            idx = int(idx)
            if t.kind == "struct":
                md = m.member_decor[(type_id, idx)]
                address += md[DEC_OFFSET][0]
                type_id = t.members[idx]
                layout = md
            elif t.kind in ("array", "runtime_array"):
                address += m.decor[type_id][DEC_ARRAY_STRIDE][0] * idx
                type_id = t.elem
            elif t.kind == "matrix":
                address += layout[DEC_MATRIX_STRIDE][0] * idx
                type_id = t.col
            elif t.kind == "vector":
                address += 4 * idx
                type_id = t.elem
            else:
                raise NotImplementedError(f"access chain into {t.kind}")
        return (kind, address, type_id, layout)

    def _ext(self, inst, x):
        b = self.b
        mp = self._map
        if inst == 4:
            return mp(lambda v: F32(abs(v)), x[0])
        if inst == 13:
            return mp(b.sin, x[0])
        if inst == 14:
            return mp(b.cos, x[0])
        if inst == 16:
            return mp(b.asin, x[0])
        if inst == 17:
            return mp(b.acos, x[0])
        if inst == 26:
            return mp(b.pow, x[0], x[1])
        if inst == 27:
            return mp(b.exp, x[0])
        if inst == 31:
            return mp(lambda v: F32(np.sqrt(v)), x[0])
        if inst == 32:  # InverseSqrt = 1 / sqrt(x), two correctly rounded operations
            return mp(lambda v: F32(1) / F32(np.sqrt(v)), x[0])
        if inst == 68:  # Cross
            p, q = x[0], x[1]
            return [p[1] * q[2] - p[2] * q[1], p[2] * q[0] - p[0] * q[2], p[0] * q[1] - p[1] * q[0]]
        if inst == 37:
            return mp(b.fmin, x[0], x[1])
        if inst == 40:
            return mp(b.fmax, x[0], x[1])
        if inst == 43:
            return mp(lambda v, lo, hi: b.fmin(b.fmax(v, lo), hi), x[0], x[1], x[2])
        if inst == 46:  # FMix: x * (1 - a) + y * a
            one = F32(1)
            return mp(lambda p, q, w: p * (one - w) + q * w, x[0], x[1], x[2])
        if inst == 49:  # SmoothStep
            def ss(e0, e1, v):
                t = b.fmin(b.fmax((v - e0) / (e1 - e0), F32(0)), F32(1))
                return t * t * (F32(3) - F32(2) * t)
            return mp(ss, x[0], x[1], x[2])
        if inst == 66:
            v = x[0]
            return F32(np.sqrt(self._dot(v, v))) if isinstance(v, list) else F32(abs(v))
        if inst == 67:
            d = mp(lambda p, q: p - q, x[0], x[1])
            return F32(np.sqrt(self._dot(d, d)))
        if inst == 69:
            v = x[0]
            inv = F32(1) / F32(np.sqrt(self._dot(v, v)))
            return [c * inv for c in v]
        raise NotImplementedError(f"GLSL.std.450 instruction {inst}")
