"""Test infrastructure: a small evaluator for the subset of GLSL that a compute shader like the reference's
deferred pass uses, so that the SHADER'S OWN TEXT can be run on the CPU (per invocation, in binary64) and its result
compared with the CPU restatement in oracle/.  Nothing under vk-renderer_amd/ uses it.

Supported: object-like #define; functions with float / int / vecN / matN / void types; declarations, assignments and
compound assignments; if / else, for, return; ?:, ||, &&, comparisons, + - * /, unary - and !, postfix ++; calls,
constructors (vec2/3/4, ivec2, float, int), swizzles, member access and indexing on host-supplied objects; the
builtins listed in BUILTINS.  Floating literals are binary32 values (GLSL's float), an operation whose operands are
both literal constants is folded to binary32 as a GLSL front end does, everything else is evaluated in binary64.
Whatever is outside the subset raises GlslError - nothing is skipped silently inside a function body."""
from __future__ import annotations

import re

import numpy as np


class GlslError(Exception):
    pass


class Const(float):
    """A value made of literals only (folds to binary32)."""


# Working precision of everything that is not a literal: binary64 by default; binary32 (every operation rounded
# to float, as a GPU evaluates GLSL's highp float) to compare operation ORDER with a binary32 restatement bit for bit.
REAL = np.float64


def set_precision(real):
    global REAL
    if real not in (np.float32, np.float64):
        raise GlslError("precision is np.float32 or np.float64")
    REAL = real


def R(x):
    return REAL(x)


TOKEN = re.compile(r"\s*(?:(\d+\.\d*(?:[eE][-+]?\d+)?[fF]?|\.\d+(?:[eE][-+]?\d+)?[fF]?|\d+[eE][-+]?\d+[fF]?)|(\d+)|([A-Za-z_]\w*)|(\+\+|--|<=|>=|==|!=|&&|\|\||[-+*/]=|[-+*/<>=!?:;,.(){}\[\]]))")
TYPES = {"float", "int", "uint", "bool", "vec2", "vec3", "vec4", "ivec2", "ivec3", "uvec2", "uvec3", "mat3", "mat4", "void"}
SWIZZLE = {c: i for i, c in enumerate("xyzw")} | {c: i for i, c in enumerate("rgba")}


def tokenize(text: str) -> list:
    out, pos = [], 0
    text = text.rstrip()
    while pos < len(text):
        m = TOKEN.match(text, pos)
        if not m:
            raise GlslError(f"cannot tokenize at {text[pos:pos + 30]!r}")
        pos = m.end()
        if m.group(1) is not None:
            out.append(("float", Const(np.float32(m.group(1).rstrip("fF")))))
        elif m.group(2) is not None:
            out.append(("int", int(m.group(2))))
        elif m.group(3) is not None:
            out.append(("name", m.group(3)))
        else:
            out.append(("op", m.group(4)))
    return out


def strip_comments(src: str) -> str:
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", src, flags=re.S))


def preprocess(src: str) -> str:
    src = strip_comments(src)
    defines, lines = {}, []
    for line in src.splitlines():
        s = line.strip()
        if s.startswith("#define"):
            parts = s.split(None, 2)
            if "(" in parts[1]:
                raise GlslError("function-like macros are outside the subset")
            defines[parts[1]] = parts[2] if len(parts) > 2 else ""
        elif s.startswith("#"):
            continue  # #version, #extension
        else:
            lines.append(line)
    body = "\n".join(lines)
    for name, value in defines.items():
        body = re.sub(rf"\b{re.escape(name)}\b", value, body)
    return body


def find_functions(src: str) -> dict:
    """name -> (return type, [(type, name)], body tokens) for every function definition at file scope."""
    funcs = {}
    for m in re.finditer(r"\b(" + "|".join(sorted(TYPES)) + r")\s+([A-Za-z_]\w*)\s*\(([^)]*)\)\s*\{", src):
        depth, i = 1, m.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        params = []
        for p in [p.strip() for p in m.group(3).split(",") if p.strip()]:
            words = [w for w in p.split() if w not in ("const", "in", "highp")]
            if len(words) != 2 or words[0] not in TYPES:
                raise GlslError(f"parameter outside the subset: {p!r}")
            params.append((words[0], words[1]))
        funcs[m.group(2)] = (m.group(1), params, tokenize(src[m.end():i - 1]))
    return funcs


# ------------------------------------------------------------------------------------------ parser -> nested tuples

class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", None)

    def take(self, kind=None, value=None):
        tok = self.peek()
        if (kind and tok[0] != kind) or (value is not None and tok[1] != value):
            raise GlslError(f"expected {value or kind}, found {tok}")
        self.i += 1
        return tok

    def at(self, value):
        return self.peek() == ("op", value)

    def block(self):
        stmts = []
        while self.peek()[0] != "eof" and not self.at("}"):
            stmts.append(self.statement())
        return ("block", stmts)

    def statement(self):
        k, v = self.peek()
        if self.at("{"):
            self.take()
            b = self.block()
            self.take("op", "}")
            return b
        if k == "name" and v == "if":
            self.take()
            self.take("op", "(")
            cond = self.expr()
            self.take("op", ")")
            then = self.statement()
            other = None
            if self.peek() == ("name", "else"):
                self.take()
                other = self.statement()
            return ("if", cond, then, other)
        if k == "name" and v == "for":
            self.take()
            self.take("op", "(")
            init = self.statement()
            cond = self.expr()
            self.take("op", ";")
            post = self.expr()
            self.take("op", ")")
            return ("for", init, cond, post, self.statement())
        if k == "name" and v == "return":
            self.take()
            e = None if self.at(";") else self.expr()
            self.take("op", ";")
            return ("return", e)
        if k == "name" and v == "const":
            self.take()
            k, v = self.peek()
        if k == "name" and v in TYPES and self.peek(1)[0] == "name":
            self.take()
            name = self.take("name")[1]
            init = None
            if self.at("="):
                self.take()
                init = self.expr()
            self.take("op", ";")
            return ("decl", v, name, init)
        e = self.expr()
        if self.peek()[0] == "op" and self.peek()[1] in ("=", "+=", "-=", "*=", "/="):
            op = self.take()[1]
            rhs = self.expr()
            self.take("op", ";")
            if e[0] != "var":
                raise GlslError("only plain variables are assigned in the subset")
            return ("assign", e[1], op, rhs)
        self.take("op", ";")
        return ("expr", e)

    def expr(self):
        cond = self.binary(0)
        if self.at("?"):
            self.take()
            a = self.expr()
            self.take("op", ":")
            return ("?:", cond, a, self.expr())
        return cond

    LEVELS = [("||",), ("&&",), ("==", "!="), ("<", ">", "<=", ">="), ("+", "-"), ("*", "/")]

    def binary(self, level):
        if level == len(self.LEVELS):
            return self.unary()
        left = self.binary(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.LEVELS[level]:
            op = self.take()[1]
            left = ("bin", op, left, self.binary(level + 1))
        return left

    def unary(self):
        if self.at("-") or self.at("!") or self.at("+"):
            op = self.take()[1]
            return ("un", op, self.unary())
        return self.postfix()

    def postfix(self):
        k, v = self.take()
        if k in ("float", "int"):
            node = ("lit", v)
        elif k == "op" and v == "(":
            node = self.expr()
            self.take("op", ")")
        elif k == "name":
            node = ("var", v)
        else:
            raise GlslError(f"unexpected token {(k, v)}")
        while True:
            if self.at("("):
                if node[0] != "var":
                    raise GlslError("call of a non-name")
                self.take()
                args = []
                while not self.at(")"):
                    args.append(self.expr())
                    if self.at(","):
                        self.take()
                self.take("op", ")")
                node = ("call", node[1], args)
            elif self.at("["):
                self.take()
                idx = self.expr()
                self.take("op", "]")
                node = ("index", node, idx)
            elif self.at("."):
                self.take()
                node = ("member", node, self.take("name")[1])
            elif self.at("++"):
                self.take()
                if node[0] != "var":
                    raise GlslError("++ on a non-variable")
                node = ("postinc", node[1])
            else:
                return node


# ------------------------------------------------------------------------------------------ evaluator

class _Return(Exception):
    def __init__(self, value):
        self.value = value


def _vec(*parts):
    flat = []
    for p in parts:
        flat.extend(np.asarray(p, REAL).reshape(-1).tolist())
    return np.array(flat, REAL)


def _ctor(n):
    def make(*args):
        v = _vec(*args)
        if v.size == 1:
            return np.full(n, v[0])
        if v.size < n:
            raise GlslError(f"vec{n} from {v.size} components")
        return v[:n].copy()
    return make


def _dot(a, b):
    a, b = np.asarray(a, REAL), np.asarray(b, REAL)
    acc = a[0] * b[0]
    for i in range(1, a.size):
        acc = acc + a[i] * b[i]  # left to right, every step rounded to the working precision
    return R(acc)


def _length(v):
    return R(np.sqrt(_dot(v, v)))


_libm = None


def _pow(x, y):
    """pow in the working precision; in binary32 the C library's powf, so that a comparison with a C restatement
    is not blurred by two implementations of the same function."""
    global _libm
    if REAL is np.float32:
        if _libm is None:
            import ctypes
            _libm = ctypes.CDLL("libm.so.6")
            _libm.powf.restype, _libm.powf.argtypes = ctypes.c_float, [ctypes.c_float, ctypes.c_float]
        one = lambda a: np.float32(_libm.powf(float(a), float(y)))
        return one(x) if np.ndim(x) == 0 else np.array([one(a) for a in np.asarray(x).reshape(-1)], np.float32).reshape(np.shape(x))
    return np.power(R(x), R(y)) if np.ndim(x) == 0 else np.power(np.asarray(x, REAL), R(y))


def _matmul(m, v):
    """matrix * vector as the sum of the columns scaled by the vector's components, left to right (and the
    same per column for matrix * matrix)."""
    m, v = np.asarray(m, REAL), np.asarray(v, REAL)
    if v.ndim == 2:
        return np.stack([_matmul(m, v[:, c]) for c in range(v.shape[1])], axis=1)
    acc = m[:, 0] * v[0]
    for c in range(1, m.shape[1]):
        acc = acc + m[:, c] * v[c]
    return acc


def _mat3(*cols):
    if len(cols) != 3 or any(np.size(c) != 3 for c in cols):
        raise GlslError("mat3 is built from three vec3 columns in the subset")
    return np.stack([np.asarray(c, REAL) for c in cols], axis=1)  # [row][column]


BUILTINS = {
    "mat3": _mat3, "nonuniformEXT": lambda x: x,
    "vec2": _ctor(2), "vec3": _ctor(3), "vec4": _ctor(4), "ivec2": lambda *a: _ctor(2)(*a).astype(np.int64),
    "float": lambda x: R(x), "int": lambda x: int(x),
    "dot": _dot, "length": _length, "normalize": lambda v: np.asarray(v, REAL) * (R(1.0) / _length(v)),  # x * inversesqrt(dot(x, x)), as compilers expand it
    "sqrt": lambda x: np.sqrt(x), "inversesqrt": lambda x: 1.0 / np.sqrt(x),
    "pow": lambda x, y: _pow(x, y),
    "max": lambda a, b: np.maximum(a, b) if np.ndim(a) or np.ndim(b) else (a if a > b else b) if not (a != a or b != b) else (b if a != a else a),
    "min": lambda a, b: np.minimum(a, b) if np.ndim(a) or np.ndim(b) else (a if a < b else b) if not (a != a or b != b) else (b if a != a else a),
    "clamp": lambda x, lo, hi: np.minimum(np.maximum(x, lo), hi),
    "mix": lambda a, b, t: np.asarray(a, REAL) * (R(1.0) - R(t)) + np.asarray(b, REAL) * R(t) if np.ndim(a) or np.ndim(b) else R(a) * (R(1.0) - R(t)) + R(b) * R(t),
    "abs": lambda x: np.abs(x), "cross": lambda a, b: np.cross(a, b),
}


class Interpreter:
    def __init__(self, source: str, globals_: dict, extra_builtins: dict):
        self.funcs = {}
        for name, (ret, params, toks) in find_functions(preprocess(source)).items():
            p = Parser(toks)
            body = p.block()
            if p.peek()[0] != "eof":
                raise GlslError(f"trailing tokens in {name}")
            self.funcs[name] = (ret, params, body)
        self.globals = globals_
        self.builtins = BUILTINS | extra_builtins

    def call(self, name, args):
        if name in self.funcs:
            ret, params, body = self.funcs[name]
            if len(params) != len(args):
                raise GlslError(f"{name}: {len(args)} arguments for {len(params)} parameters")
            scope = {pn: (R(a) if pt == "float" and np.ndim(a) == 0 else a) for (pt, pn), a in zip(params, args)}
            try:
                self.run(body, [scope])
            except _Return as r:
                return r.value
            return None
        if name in self.builtins:
            return self.builtins[name](*[R(a) if isinstance(a, (Const, float)) and not isinstance(a, bool) else a for a in args])
        raise GlslError(f"no function {name}")

    def lookup(self, name, scopes):
        for s in reversed(scopes):
            if name in s:
                return s[name]
        if name in self.globals:
            return self.globals[name]
        raise GlslError(f"unknown name {name}")

    def run(self, node, scopes):
        kind = node[0]
        if kind == "block":
            scopes = scopes + [{}]
            for s in node[1]:
                self.run(s, scopes)
        elif kind == "decl":
            v = self.eval(node[3], scopes) if node[3] is not None else None
            if node[1] == "float" and v is not None:
                v = R(v)
            elif node[1] == "int" and v is not None:
                v = int(v)
            scopes[-1][node[2]] = v
        elif kind == "assign":
            rhs = self.eval(node[3], scopes)
            for s in reversed(scopes):
                if node[1] in s:
                    cur = s[node[1]]
                    if node[2] != "=":
                        rhs = self.arith(node[2][0], cur, rhs)
                    s[node[1]] = R(rhs) if isinstance(rhs, Const) else rhs
                    return
            if node[1] in self.globals and node[2] == "=":  # an `out` variable or a built-in output
                self.globals[node[1]] = R(rhs) if isinstance(rhs, Const) else rhs
                return
            raise GlslError(f"assignment to unknown variable {node[1]}")
        elif kind == "if":
            if self.eval(node[1], scopes):
                self.run(node[2], scopes)
            elif node[3] is not None:
                self.run(node[3], scopes)
        elif kind == "for":
            scopes = scopes + [{}]
            self.run(node[1], scopes)
            while self.eval(node[2], scopes):
                self.run(node[4], scopes)
                self.eval(node[3], scopes)
        elif kind == "return":
            raise _Return(self.eval(node[1], scopes) if node[1] is not None else None)
        elif kind == "expr":
            self.eval(node[1], scopes)
        else:
            raise GlslError(f"statement {kind}")

    @staticmethod
    def arith(op, a, b):
        both_const = isinstance(a, (Const, int)) and isinstance(b, (Const, int)) and not (isinstance(a, int) and isinstance(b, int))
        if isinstance(a, int) and isinstance(b, int) and not isinstance(a, bool) and not isinstance(b, bool):
            if op == "/":
                raise GlslError("integer division is outside the subset")
            return {"+": a + b, "-": a - b, "*": a * b}[op]
        x, y = (a, b)
        if op == "*" and np.ndim(x) == 2 and np.ndim(y) in (1, 2):
            r = _matmul(x, y)
        elif np.ndim(x) == 0 and np.ndim(y) == 0:
            x, y = R(x), R(y)
            r = {"+": lambda: x + y, "-": lambda: x - y, "*": lambda: x * y, "/": lambda: x / y}[op]()
        else:
            x = np.asarray(x, REAL) if np.ndim(x) else R(x)
            y = np.asarray(y, REAL) if np.ndim(y) else R(y)
            r = {"+": lambda: x + y, "-": lambda: x - y, "*": lambda: x * y, "/": lambda: x / y}[op]()
        if both_const:
            return Const(np.float32(r))  # constant folding in the language's own precision
        return r

    def eval(self, node, scopes):
        kind = node[0]
        if kind == "lit":
            return node[1]
        if kind == "var":
            return self.lookup(node[1], scopes)
        if kind == "bin":
            op = node[1]
            if op == "&&":
                return bool(self.eval(node[2], scopes)) and bool(self.eval(node[3], scopes))
            if op == "||":
                return bool(self.eval(node[2], scopes)) or bool(self.eval(node[3], scopes))
            a, b = self.eval(node[2], scopes), self.eval(node[3], scopes)
            if op in "+-*/":
                return self.arith(op, a, b)
            a, b = float(a), float(b)
            return {"<": a < b, ">": a > b, "<=": a <= b, ">=": a >= b, "==": a == b, "!=": a != b}[op]
        if kind == "un":
            v = self.eval(node[2], scopes)
            if node[1] == "!":
                return not v
            if node[1] == "+":
                return v
            return Const(-float(v)) if isinstance(v, Const) else -v
        if kind == "?:":
            return self.eval(node[2], scopes) if self.eval(node[1], scopes) else self.eval(node[3], scopes)
        if kind == "call":
            return self.call(node[1], [self.eval(a, scopes) for a in node[2]])
        if kind == "index":
            return self.eval(node[1], scopes)[int(self.eval(node[2], scopes))]
        if kind == "member":
            base = self.eval(node[1], scopes)
            name = node[2]
            if isinstance(base, dict):
                return base[name]
            if all(c in SWIZZLE for c in name):
                arr = np.asarray(base)
                picked = [arr[SWIZZLE[c]] for c in name]
                return R(picked[0]) if len(picked) == 1 else np.array(picked, arr.dtype if arr.dtype.kind == "i" else REAL)
            raise GlslError(f"member {name}")
        if kind == "postinc":
            for s in reversed(scopes):
                if node[1] in s:
                    old = s[node[1]]
                    s[node[1]] = old + 1
                    return old
            raise GlslError(f"++ on unknown variable {node[1]}")
        raise GlslError(f"expression {kind}")


def run_deferred_shader(source: str, gbuf: dict, lights: np.ndarray, push: np.ndarray, width: int, height: int) -> np.ndarray:
    """Runs `main()` of a deferred-resolve compute shader once per pixel.  gbuf: albedo / material (H,W,4) uint8,
    normal (H,W,4) float16, depth (H,W) float32 - sampled as the reference's nearest-filtered attachments are, at
    texel centres.  lights: (L,12) float32 std140 records; push: 20 floats (cameraPos, pointLightRadius, inverseVP
    column-major).  Returns what the shader passes to imageStore, (H,W,4) float64."""
    planes = [gbuf["albedo"].astype(REAL) / REAL(255.0), gbuf["material"].astype(REAL) / REAL(255.0),
              gbuf["normal"].astype(REAL), np.repeat(gbuf["depth"].astype(REAL)[..., None], 4, axis=-1)]
    lights = np.asarray(lights, np.float32).reshape(-1, 12).astype(REAL)
    records = [{"pos": r[0:3], "opening_angle": r[3], "color": r[4:7], "penumbra_angle": r[7], "dir": r[8:11], "_pad": r[11]} for r in lights]
    push = np.asarray(push, np.float32).astype(REAL)
    out = np.zeros((height, width, 4), REAL)

    def texture(sampler, uv):
        x = min(max(int(np.floor(uv[0] * width)), 0), width - 1)
        y = min(max(int(np.floor(uv[1] * height)), 0), height - 1)
        return planes[sampler][y, x].copy()

    def image_store(image, coord, value):
        out[int(coord[1]), int(coord[0])] = value

    g = {"samplers": [0, 1, 2, 3], "outTex": "outTex",
         "push": {"cameraPos": push[0:3], "pointLightRadius": push[3], "inverseVP": push[4:20].reshape(4, 4).T},  # [row][col] from column-major
         "ubo": {"lights": records},
         "gl_NumWorkGroups": np.array([width, height, 1], REAL), "gl_WorkGroupSize": np.array([1, 1, 1], REAL)}
    interp = Interpreter(source, g, {"texture": texture, "imageStore": image_store})
    for y in range(height):
        for x in range(width):
            g["gl_GlobalInvocationID"] = np.array([x, y, 0], REAL)
            interp.call("main", [])
    return out


def run_offscreen_shaders(vert_source: str, frag_source: str, normal_obj, tangent_obj, matrices, matrix_index, albedo_texel, specular_texel,
                          normal_texel) -> dict:
    """The reference's G-buffer pass for one surface sample per pixel: its vertex shader's main() with the pixel's
    own attributes (object-space normal / tangent, model matrix index), then its fragment shader's main() with the
    frame that came out and the pixel's three texels.  Inputs (H,W,3|4) / (H,W) / (K,16 column-major); returns the
    three colour outputs as (H,W,4) binary64."""
    H, W = matrix_index.shape
    mats = [np.asarray(m, np.float64).reshape(4, 4).T for m in np.asarray(matrices).reshape(-1, 16)]  # [row][column]
    out = {k: np.zeros((H, W, 4), np.float64) for k in ("outColor", "outSpecular", "outNormal")}
    vg = {"push": {"vp": np.eye(4)}, "gl_DrawID": 0, "gl_InstanceIndex": 0, "gl_Position": None,
          "ubo0": {"modelData": [{"matIdx": 0, "modelMatrixBaseIndex": 0}]}, "ubo1": {"modelMatrices": mats},
          "inPos": np.zeros(3), "inUV": np.zeros(3), "outUV": None, "outTexOffset": None, "outTBN": None}
    vert = Interpreter(vert_source, vg, {})
    texels = {}
    fg = {"samplers": [0, 1, 2], "inUV": np.zeros(2), "inTexOffset": 0, "inTBN": None, "outColor": None, "outSpecular": None, "outNormal": None}
    frag = Interpreter(frag_source, fg, {"texture": lambda sampler, uv: texels[int(sampler)].copy()})
    for y in range(H):
        for x in range(W):
            vg["ubo0"]["modelData"][0]["modelMatrixBaseIndex"] = int(matrix_index[y, x])
            vg["inNormal"] = np.asarray(normal_obj[y, x, :3], np.float64)
            vg["inTangent"] = np.asarray(tangent_obj[y, x, :3], np.float64)
            vert.call("main", [])
            fg["inTBN"] = vg["outTBN"]
            fg["inTexOffset"] = vg["outTexOffset"]
            texels.update({0: np.asarray(albedo_texel[y, x], np.float64), 1: np.asarray(normal_texel[y, x], np.float64),
                           2: np.asarray(specular_texel[y, x], np.float64)})
            frag.call("main", [])
            for k in out:
                out[k][y, x] = fg[k]
    return out
