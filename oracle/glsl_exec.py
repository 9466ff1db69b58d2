"""TEST INFRASTRUCTURE ONLY (never imported by the product path): a small interpreter for the GLSL subset the
reference's screen-space fragment shaders are written in, used in THIS container to execute the reference's own
shader text (read from /root/reference/renderer/shaders at run time, never copied into the repo) on synthetic inputs.
Its outputs are committed as fixtures (tests/golden/make_glsl_golden.py) and pin the hand-written oracles
(oracle/mil_oracle.py, oracle/mil_oracle.c) against the reference's source rather than against a restatement of it.

Execution model: SIMT over all fragments at once - every run-time value is a float32 NumPy array (P, k), k = 1 for
float, 2..4 for vectors; uniforms have P = 1 and broadcast.  One IEEE float32 operation per GLSL operator, in source
order, no contraction.  What the GLSL / Vulkan specifications leave to the implementation is fixed here the way the C
oracle fixes it, and listed so that it is not mistaken for something the reference pins:
  * constant expressions of literals / const-qualified values fold in double and are narrowed when they meet a
    run-time value (glslang's front end does this);
  * dot() and mat * vec sum left to right; normalize(v) = v * (1 / sqrt(dot(v, v))); mix(a, b, t) = a*(1-t) + b*t;
    pow(x, n) with a small integer constant n is repeated multiplication, otherwise float32 powf; clamp = min(max());
    exp / pow / acos / sqrt / inversesqrt are the correctly rounded float32 of the double result;
  * subpassLoad / texture() return whatever decoded texels the caller supplies (format decode is the driver's).
Control flow: `for` bounds and `if` conditions may be per-fragment; both branches of an `if` are executed under a mask
and assignments merge by the mask (side-effect free shaders only), early `return` inside a divergent `if` is merged the
same way.
"""
import os
import re

import numpy as np

F32 = np.float32
VEC = {'float': 1, 'vec2': 2, 'vec3': 3, 'vec4': 4}
SWZ = {c: i for s in ('xyzw', 'rgba', 'stpq') for i, c in enumerate(s)}


# --------------------------------------------------------------------------------------------- source handling
def preprocess(path, seen=None, macros=None):
    """#include expansion (the reference uses vulkano-shaders' textual include), comments stripped, object-like
    #define macros substituted, #ifdef / #ifndef / #endif honoured."""
    seen = seen if seen is not None else set()
    macros = macros if macros is not None else {}
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', '', src)
    out, live = [], [True]
    for line in src.split('\n'):
        m = re.match(r'\s*#\s*(\w+)\s*(.*)', line)
        if m:
            d, rest = m.group(1), m.group(2).strip()
            if d in ('ifdef', 'ifndef'):
                live.append(live[-1] and ((rest in macros) == (d == 'ifdef')))
            elif d == 'endif':
                live.pop()
            elif not live[-1]:
                pass
            elif d == 'include':
                inc = os.path.join(os.path.dirname(path), rest.strip('"'))
                if inc not in seen:
                    seen.add(inc)
                    out.append(preprocess(inc, seen, macros))
            elif d == 'define':
                name, _, body = rest.partition(' ')
                macros[name] = body.strip()
            elif d not in ('version', 'extension'):
                raise SyntaxError(f'preprocessor directive #{d} is not in the subset')
            continue
        if live[-1]:
            for _ in range(4):                                     # nested object-like macros
                line = re.sub(r'\b(' + '|'.join(map(re.escape, macros)) + r')\b', lambda mm: macros[mm.group(1)], line) if macros else line
            out.append(line)
    return '\n'.join(out)


TOK = re.compile(r'\s*(?:(\d+\.\d*(?:[eE][+-]?\d+)?[fF]?|\.\d+(?:[eE][+-]?\d+)?[fF]?|\d+[eE][+-]?\d+[fF]?)|(\d+)([uU]?)|([A-Za-z_]\w*)|'
                 r'(\+\+|--|\+=|-=|\*=|/=|==|!=|<=|>=|&&|\|\||[-+*/<>=!?:;,.(){}\[\]]))')


def tokenize(src):
    toks, i = [], 0
    src = src.rstrip()
    while i < len(src):
        m = TOK.match(src, i)
        if not m:
            raise SyntaxError(f'cannot tokenize at {src[i:i + 30]!r}')
        i = m.end()
        if m.group(1):
            toks.append(('f', float(m.group(1).rstrip('fF'))))
        elif m.group(2):
            toks.append(('i', int(m.group(2))))
        elif m.group(4):
            toks.append(('id', m.group(4)))
        else:
            toks.append(('op', m.group(5)))
    return toks


QUALS = {'const', 'in', 'out', 'inout', 'highp', 'mediump', 'lowp', 'flat'}
TYPES = {'void', 'float', 'int', 'uint', 'bool', 'vec2', 'vec3', 'vec4', 'mat4', 'mat3', 'subpassInput', 'sampler2D'}


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0
        self.structs, self.funcs, self.globals, self.blocks, self.ins, self.outs, self.samplers = {}, {}, [], {}, [], [], []

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ('eof', None)

    def next(self):
        tok = self.peek()
        self.i += 1
        return tok

    def accept(self, val):
        if self.peek()[1] == val and self.peek()[0] in ('op', 'id'):
            self.i += 1
            return True
        return False

    def expect(self, val):
        if not self.accept(val):
            raise SyntaxError(f'expected {val!r}, got {self.peek()} at token {self.i}')

    def is_type(self, tok):
        return tok[0] == 'id' and (tok[1] in TYPES or tok[1] in self.structs)

    # ---- top level
    def parse(self):
        while self.peek()[0] != 'eof':
            self.toplevel()
        return self

    def skip_layout(self):
        if self.accept('layout'):
            self.expect('(')
            depth = 1
            while depth:
                tok = self.next()[1]
                depth += (tok == '(') - (tok == ')')

    def toplevel(self):
        if self.accept(';'):
            return
        if self.accept('struct'):
            name = self.next()[1]
            self.structs[name] = self.members()
            self.expect(';')
            return
        self.skip_layout()
        quals = set()
        while self.peek()[1] in QUALS | {'uniform'}:
            quals.add(self.next()[1])
        ty = self.next()[1]
        if 'uniform' in quals and self.peek()[1] == '{':          # uniform block: Block { members } instance;
            mem = self.members()
            inst = self.next()[1] if self.peek()[0] == 'id' else None
            self.expect(';')
            self.blocks[inst or ty] = (mem, inst is not None)
            return
        name = self.next()[1]
        if self.accept('('):                                        # function
            params = []
            while not self.accept(')'):
                pq = set()
                while self.peek()[1] in QUALS:
                    pq.add(self.next()[1])
                pty = self.next()[1]
                if pty == 'void':
                    continue
                params.append((pty, self.next()[1], pq))
                self.accept(',')
            self.funcs[name] = (ty, params, self.block())
            return
        if 'uniform' in quals:
            self.samplers.append((ty, name))
            self.expect(';')
            return
        if 'in' in quals or 'out' in quals:
            (self.ins if 'in' in quals else self.outs).append((ty, name))
            self.expect(';')
            return
        init = None
        if self.accept('='):
            init = self.expr()
        self.expect(';')
        self.globals.append((ty, name, init, 'const' in quals))

    def members(self):
        self.expect('{')
        mem = []
        while not self.accept('}'):
            ty = self.next()[1]
            name = self.next()[1]
            n = None
            if self.accept('['):
                n = self.expr()
                self.expect(']')
            self.expect(';')
            mem.append((ty, name, n))
        return mem

    # ---- statements
    def block(self):
        self.expect('{')
        body = []
        while not self.accept('}'):
            body.append(self.stmt())
        return ('block', body)

    def stmt(self):
        tok = self.peek()
        if tok[1] == '{':
            return self.block()
        if self.accept('return'):
            e = None if self.peek()[1] == ';' else self.expr()
            self.expect(';')
            return ('return', e)
        if self.accept('if'):
            self.expect('(')
            c = self.expr()
            self.expect(')')
            a = self.stmt()
            b = self.stmt() if self.accept('else') else None
            return ('if', c, a, b)
        if self.accept('for'):
            self.expect('(')
            init = self.simple()
            self.expect(';')
            cond = self.expr()
            self.expect(';')
            inc = self.simple()
            self.expect(')')
            return ('for', init, cond, inc, self.stmt())
        if self.accept('discard'):
            self.expect(';')
            return ('discard',)
        s = self.simple()
        self.expect(';')
        return s

    def simple(self):
        const = False
        while self.peek()[1] in QUALS:
            const |= self.next()[1] == 'const'
        if self.is_type(self.peek()) and self.peek(1)[0] == 'id':
            ty = self.next()[1]
            decls = []
            while True:
                name = self.next()[1]
                init = self.expr() if self.accept('=') else None
                decls.append(('decl', ty, name, init, const))
                if not self.accept(','):
                    break
            return decls[0] if len(decls) == 1 else ('block_noscope', decls)
        e = self.expr()
        if self.peek()[1] in ('=', '+=', '-=', '*=', '/='):
            op = self.next()[1]
            return ('assign', op, e, self.expr())
        return ('expr', e)

    # ---- expressions (precedence climbing)
    def expr(self):
        c = self.binary(0)
        if self.accept('?'):
            a = self.expr()
            self.expect(':')
            return ('cond', c, a, self.expr())
        return c

    PREC = [('||',), ('&&',), ('==', '!='), ('<', '>', '<=', '>='), ('+', '-'), ('*', '/')]

    def binary(self, lvl):
        if lvl == len(self.PREC):
            return self.unary()
        e = self.binary(lvl + 1)
        while self.peek()[0] == 'op' and self.peek()[1] in self.PREC[lvl]:
            op = self.next()[1]
            e = ('bin', op, e, self.binary(lvl + 1))
        return e

    def unary(self):
        if self.accept('-'):
            return ('neg', self.unary())
        if self.accept('+'):
            return self.unary()
        if self.accept('!'):
            return ('not', self.unary())
        if self.accept('++') or self.accept('--'):
            raise SyntaxError('prefix increment is not in the subset')
        return self.postfix()

    def postfix(self):
        tok = self.next()
        if tok[0] == 'f':
            e = ('const', float(tok[1]), 'float')
        elif tok[0] == 'i':
            e = ('const', int(tok[1]), 'int')
        elif tok[1] == '(':
            e = self.expr()
            self.expect(')')
        elif tok[0] == 'id':
            if self.peek()[1] == '(':
                self.next()
                args = []
                while not self.accept(')'):
                    args.append(self.expr())
                    self.accept(',')
                e = ('call', tok[1], args)
            elif tok[1] in ('true', 'false'):
                e = ('const', tok[1] == 'true', 'bool')
            else:
                e = ('var', tok[1])
        else:
            raise SyntaxError(f'unexpected token {tok}')
        while True:
            if self.accept('.'):
                e = ('member', e, self.next()[1])
            elif self.accept('['):
                e = ('index', e, self.expr())
                self.expect(']')
            elif self.peek()[1] in ('++', '--') and self.peek()[0] == 'op':
                op = self.next()[1]
                e = ('postinc', e, 1 if op == '++' else -1)
            else:
                return e


# --------------------------------------------------------------------------------------------- values
class Const:
    """A compile-time constant: folds in double (glslang), narrows to float32 on contact with run-time data."""
    __slots__ = ('v', 'k')

    def __init__(self, v, k=1):
        self.v, self.k = np.asarray(v, np.float64).reshape(-1) if k else v, k     # k = 0: integer (python int)

    def rt(self):
        return np.asarray(self.v, F32).reshape(1, -1) if self.k else F32(self.v).reshape(1, 1)


def is_int(x):
    return isinstance(x, (int, np.integer)) and not isinstance(x, bool)


def to_rt(x):
    """run-time float32 array (P, k)"""
    if isinstance(x, Const):
        return x.rt()
    if is_int(x):
        return np.full((1, 1), x, F32)
    return x


class Ret(Exception):
    pass


class Shader:
    def __init__(self, path):
        self.p = Parser(tokenize(preprocess(path))).parse()

    # ---- run main() over P fragments
    def run(self, n_frag, uniforms=None, frag_coord=None, inputs=None, samplers=None, varyings=None):
        """uniforms: {block instance or member name: {member: value}}; inputs: {subpassInput name: (P, 4) float32};
        samplers: {name: callable(uv (P, 2)) -> (P, 4)}; returns {out name: (P, k)}."""
        self.P = n_frag
        self.inputs, self.samplers = inputs or {}, samplers or {}
        g = {}
        for ty, name, init, const in self.p.globals:
            v = self.eval(init, [g]) if init is not None else None
            g[name] = v if const else to_rt(v)
        for inst, (mem, named) in self.p.blocks.items():
            vals = (uniforms or {})[inst]
            blk = {m[1]: self.uniform_value(m[0], vals[m[1]], m[2], [g]) for m in mem if m[1] in vals}
            if named:
                g[inst] = blk
            else:
                g.update(blk)
        for ty, name in self.p.samplers:
            if name in self.inputs:
                g[name] = np.asarray(self.inputs[name], F32)
            elif name in self.samplers:
                g[name] = self.samplers[name]
        if frag_coord is not None:
            g['gl_FragCoord'] = np.asarray(frag_coord, F32)
        for ty, name in self.p.ins:
            g[name] = np.asarray(varyings[name], F32)
        for ty, name in self.p.outs:
            g[name] = np.zeros((n_frag, VEC[ty]), F32)
        self.mask = None
        self.g = g
        self.exec_block(self.p.funcs['main'][2], [g, {}])
        return {name: np.broadcast_to(g[name], (n_frag, VEC[ty])).copy() for ty, name in self.p.outs}

    def uniform_value(self, ty, val, n, env):
        if n is not None:
            cnt = self.eval(n, env)
            cnt = cnt.v if isinstance(cnt, Const) else cnt
            return [self.uniform_value(ty, v, None, env) for v in list(val)[:int(cnt)]]
        if ty in self.p.structs:
            return {m[1]: self.uniform_value(m[0], val[m[1]], m[2], env) for m in self.p.structs[ty] if m[1] in val}
        if ty in ('uint', 'int'):
            return int(val)
        if ty == 'mat4':
            return ('mat4', np.asarray(val, F32).reshape(4, 4))          # [column][row]
        return np.asarray(val, F32).reshape(1, VEC[ty])

    # ---- statements
    def exec_block(self, node, env, scope=True):
        if scope:
            env = env + [{}]
        for s in node[1]:
            self.exec(s, env)

    def lookup(self, name, env):
        for e in reversed(env):
            if name in e:
                return e
        raise NameError(name)

    def exec(self, s, env):
        k = s[0]
        if k == 'block':
            self.exec_block(s, env)
        elif k == 'block_noscope':
            for d in s[1]:
                self.exec(d, env)
        elif k == 'decl':
            _, ty, name, init, const = s
            v = self.eval(init, env) if init is not None else None
            if v is not None and ty in VEC and not (const and isinstance(v, Const)):
                v = to_rt(v)
                if v.shape[1] == 1 and VEC[ty] > 1:
                    v = np.repeat(v, VEC[ty], 1)
            if v is None and ty in VEC:
                v = np.zeros((1, VEC[ty]), F32)
            if ty in ('int', 'uint') and isinstance(v, Const):
                v = int(v.v)
            env[-1][name] = v
        elif k == 'assign':
            self.assign(s[2], s[1], self.eval(s[3], env), env)
        elif k == 'expr':
            self.eval(s[1], env)
        elif k == 'return':
            self.ret_val = self.eval(s[1], env) if s[1] is not None else None
            raise Ret()
        elif k == 'for':
            env = env + [{}]
            self.exec(s[1], env)
            while True:
                c = self.eval(s[2], env)
                c = np.asarray(c).reshape(-1)
                if c.size != 1:
                    raise NotImplementedError('per-fragment loop bound')
                if not bool(c[0]):
                    break
                self.exec(s[4], env)
                self.exec(s[3], env)
        elif k == 'if':
            c = self.eval(s[1], env)
            c = np.asarray(c)
            if c.size == 1:                                        # uniform branch
                branch = s[2] if bool(c.reshape(-1)[0]) else s[3]
                if branch is not None:
                    self.exec(branch, env)
                return
            c = np.broadcast_to(c.reshape(-1, 1), (self.P, 1))
            outer = self.mask
            for cond, branch in ((c, s[2]), (~c, s[3])):
                if branch is None:
                    continue
                self.mask = cond if outer is None else (outer & cond)
                if self.mask.any():
                    self.exec(branch, env)
            self.mask = outer
        elif k == 'discard':
            raise NotImplementedError('discard')
        else:
            raise NotImplementedError(k)

    def merge(self, old, new):
        """assignment under a divergent mask"""
        new = to_rt(new)
        if self.mask is None or old is None:
            return new
        old = to_rt(old)
        return np.where(self.mask, np.broadcast_to(new, (self.P, new.shape[1])), np.broadcast_to(old, (self.P, old.shape[1]))).astype(F32)

    def assign(self, target, op, val, env):
        if op != '=':
            val = self.arith(op[0], self.eval(target, env), val)
        if target[0] == 'var':
            e = self.lookup(target[1], env)
            cur = e[target[1]]
            if is_int(cur) or (cur is None and is_int(val)):
                e[target[1]] = int(val.v) if isinstance(val, Const) else val
                return
            v = to_rt(val)
            if cur is not None and not isinstance(cur, Const) and hasattr(cur, 'shape') and v.shape[1] == 1 and cur.shape[1] > 1:
                v = np.repeat(v, cur.shape[1], 1)
            e[target[1]] = self.merge(cur, v)
        elif target[0] == 'member':                                # swizzle store: v.xyz = ..., v.w = ...
            base = to_rt(self.eval(target[1], env))
            idx = [SWZ[c] for c in target[2]]
            v = to_rt(val)
            out = np.broadcast_to(base, (max(base.shape[0], v.shape[0], self.P if self.mask is not None else 1), base.shape[1])).copy()
            out[:, idx] = np.broadcast_to(v, (out.shape[0], len(idx)))
            self.assign(target[1], '=', out, env)
        else:
            raise NotImplementedError(f'assignment to {target[0]}')

    # ---- expressions
    def arith(self, op, a, b):
        if is_int(a) and is_int(b):
            return {'+': a + b, '-': a - b, '*': a * b, '/': a // b if b else 0}[op]
        if isinstance(a, tuple) or isinstance(b, tuple):           # matrix products
            return self.matmul(a, b, op)
        if (isinstance(a, Const) or is_int(a)) and (isinstance(b, Const) or is_int(b)):       # constant folding in double
            x = a.v if isinstance(a, Const) else float(a)
            y = b.v if isinstance(b, Const) else float(b)
            with np.errstate(all='ignore'):
                return Const({'+': np.add, '-': np.subtract, '*': np.multiply, '/': np.divide}[op](x, y), 1)
        x, y = to_rt(a), to_rt(b)
        with np.errstate(all='ignore'):
            return {'+': np.add, '-': np.subtract, '*': np.multiply, '/': np.divide}[op](x, y).astype(F32)

    def matmul(self, a, b, op):
        assert op == '*'
        if isinstance(a, tuple) and not isinstance(b, tuple):      # mat4 * vec4: rows summed left to right
            m, v = a[1], to_rt(b)
            cols = [v[:, c:c + 1] for c in range(4)]
            with np.errstate(all='ignore'):
                return np.concatenate([((m[0, r] * cols[0] + m[1, r] * cols[1]) + m[2, r] * cols[2]) + m[3, r] * cols[3] for r in range(4)], 1).astype(F32)
        raise NotImplementedError('matrix product form')

    def eval(self, e, env):
        k = e[0]
        if k == 'const':
            return e[1] if e[2] in ('int', 'bool') else Const(e[1])
        if k == 'var':
            return self.lookup(e[1], env)[e[1]]
        if k == 'neg':
            v = self.eval(e[1], env)
            if is_int(v):
                return -v
            return Const(-v.v) if isinstance(v, Const) else (-v).astype(F32)
        if k == 'not':
            return ~np.asarray(self.eval(e[1], env), bool)
        if k == 'bin':
            op = e[1]
            a, b = self.eval(e[2], env), self.eval(e[3], env)
            if op in ('+', '-', '*', '/'):
                return self.arith(op, a, b)
            if op in ('&&', '||'):
                a, b = np.asarray(a, bool), np.asarray(b, bool)
                return (a & b) if op == '&&' else (a | b)
            if is_int(a) and is_int(b):
                x, y = a, b
            else:
                x, y = to_rt(a), to_rt(b)
            return {'<': np.less, '>': np.greater, '<=': np.less_equal, '>=': np.greater_equal, '==': np.equal, '!=': np.not_equal}[op](x, y)
        if k == 'cond':
            c = np.asarray(self.eval(e[1], env), bool)
            a, b = to_rt(self.eval(e[2], env)), to_rt(self.eval(e[3], env))
            return np.where(c.reshape(-1, 1), a, b).astype(F32)
        if k == 'member':
            base = self.eval(e[1], env)
            if isinstance(base, dict):
                return base[e[2]]
            idx = [SWZ[c] for c in e[2]]
            if isinstance(base, Const):
                return Const(base.v[idx])
            return base[:, idx]
        if k == 'index':
            base, i = self.eval(e[1], env), self.eval(e[2], env)
            i = int(i.v) if isinstance(i, Const) else int(i)
            if isinstance(base, list):
                return base[i]
            return to_rt(base)[:, i:i + 1]
        if k == 'postinc':
            cur = self.eval(e[1], env)
            self.lookup(e[1][1], env)[e[1][1]] = cur + e[2]
            return cur
        if k == 'call':
            return self.call(e[1], [self.eval(a, env) for a in e[2]], e[2], env)
        raise NotImplementedError(k)

    # ---- calls: constructors, built-ins, user functions
    def call(self, name, args, arg_nodes, env):
        if name in VEC and name != 'float':
            n = VEC[name]
            if all(isinstance(a, Const) or is_int(a) for a in args):
                flat = np.concatenate([(a.v if isinstance(a, Const) else np.array([float(a)])) for a in args])
                return Const(np.repeat(flat, n) if flat.size == 1 else flat[:n])
            parts = [to_rt(a) for a in args]
            rows = max(p.shape[0] for p in parts)
            v = np.concatenate([np.broadcast_to(p, (rows, p.shape[1])) for p in parts], 1)
            return (np.repeat(v, n, 1) if v.shape[1] == 1 else v[:, :n]).astype(F32)
        if name == 'float':
            a = args[0]
            return Const(float(a)) if is_int(a) else a
        if name in ('uint', 'int'):
            a = args[0]
            return int(a.v) if isinstance(a, Const) else a
        b = getattr(self, 'b_' + name, None)
        if b is not None and name not in self.p.funcs:
            return b(*args)
        ret_ty, params, body = self.p.funcs[name]
        local = {}
        for (pty, pname, pq), a in zip(params, args):
            if 'out' in pq and 'inout' not in pq:
                a = None if pty not in VEC else np.zeros((1, VEC[pty]), F32)
            local[pname] = a if (a is None or callable(a) or isinstance(a, (dict, list, tuple)) or is_int(a)) else to_rt(a)
            if pty in VEC and hasattr(local[pname], 'shape') and local[pname].shape[1] == 1 and VEC[pty] > 1:
                local[pname] = np.repeat(local[pname], VEC[pty], 1)
            if is_int(local[pname]) and pty in VEC:
                local[pname] = to_rt(local[pname])
        saved = self.mask
        try:
            self.ret_val = None
            self.exec_block(body, [env[0], local])
        except Ret:
            pass
        if self.mask is not saved:
            raise NotImplementedError('return inside a divergent branch of a called function')
        ret = self.ret_val
        for (pty, pname, pq), node in zip(params, arg_nodes):     # copy-out of out / inout parameters
            if 'out' in pq or 'inout' in pq:
                self.assign(node, '=', local[pname], env)
        return ret

    def call_function(self, name, *args):
        """Run one function of the shader on its own (e.g. PositionFromDepth); uniforms as set by the last run()."""
        return self.call(name, [np.asarray(a, F32) if not is_int(a) else a for a in args], [None] * len(args), [self.g])

    # built-ins (conventions: module docstring)
    @staticmethod
    def _f(x):
        return to_rt(x)

    def b_subpassLoad(self, name_val):
        return name_val

    def b_dot(self, a, b):
        a, b = self._f(a), self._f(b)
        with np.errstate(all='ignore'):
            acc = a[:, 0:1] * b[:, 0:1]
            for i in range(1, max(a.shape[1], b.shape[1])):
                acc = (acc + a[:, i:i + 1] * b[:, i:i + 1]).astype(F32)
        return acc.astype(F32)

    def b_normalize(self, v):
        v = self._f(v)
        with np.errstate(all='ignore'):
            inv = (F32(1.0) / np.sqrt(self.b_dot(v, v))).astype(F32)
            return (v * inv).astype(F32)

    def b_length(self, v):
        return np.sqrt(self.b_dot(v, v)).astype(F32)

    def b_sqrt(self, x):
        if isinstance(x, Const):
            return Const(np.sqrt(x.v))
        with np.errstate(all='ignore'):
            return np.sqrt(self._f(x)).astype(F32)

    def b_inversesqrt(self, x):
        with np.errstate(all='ignore'):
            return (F32(1.0) / np.sqrt(self._f(x))).astype(F32)

    def _minmax(self, fn, a, b):
        if (isinstance(a, Const) or is_int(a)) and (isinstance(b, Const) or is_int(b)):
            return Const(fn(Const(a).v if not isinstance(a, Const) else a.v, Const(b).v if not isinstance(b, Const) else b.v))
        return fn(self._f(a), self._f(b)).astype(F32)

    def b_max(self, a, b):
        return self._minmax(np.fmax, a, b)

    def b_min(self, a, b):
        return self._minmax(np.fmin, a, b)

    def b_clamp(self, x, lo, hi):
        return self.b_min(self.b_max(x, lo), hi)

    def b_abs(self, x):
        return np.abs(self._f(x)).astype(F32)

    def b_floor(self, x):
        return np.floor(self._f(x)).astype(F32)

    def b_mix(self, a, b, t):
        one_minus = self.arith('-', Const(1.0), t)
        return self.arith('+', self.arith('*', a, one_minus), self.arith('*', b, t))

    def b_pow(self, x, n):
        nv = n.v if isinstance(n, Const) else n
        if (isinstance(n, Const) or is_int(n)) and float(np.asarray(nv).reshape(-1)[0]) == int(np.asarray(nv).reshape(-1)[0]) and 1 <= int(np.asarray(nv).reshape(-1)[0]) <= 8:
            k = int(np.asarray(nv).reshape(-1)[0])              # square-and-multiply, as a compiler expands pow(x, 5)
            x = self._f(x)
            with np.errstate(all='ignore'):
                x2 = (x * x).astype(F32)
                if k == 1:
                    return x
                if k == 2:
                    return x2
                x4 = (x2 * x2).astype(F32)
                return {3: (x2 * x), 4: x4, 5: (x4 * x), 6: (x4 * x2), 7: ((x4 * x2).astype(F32) * x), 8: (x4 * x4)}[k].astype(F32)
        with np.errstate(all='ignore'):
            return np.power(self._f(x).astype(np.float64), self._f(n).astype(np.float64)).astype(F32)

    def b_exp(self, x):
        with np.errstate(all='ignore'):
            return np.exp(self._f(x).astype(np.float64)).astype(F32)

    def b_acos(self, x):
        with np.errstate(all='ignore'):
            return np.arccos(self._f(x).astype(np.float64)).astype(F32)

    def b_cos(self, x):
        return np.cos(self._f(x).astype(np.float64)).astype(F32)

    def b_texture(self, sampler, uv):
        return np.asarray(sampler(self._f(uv)), F32)

    def b_textureSize(self, sampler, lod):
        return sampler.size
