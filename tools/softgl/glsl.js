/*
 * glsl.js -- a small GLSL ES 1.00 front end + evaluator (TEST TOOLING, build container only).
 *
 * Purpose: execute the reference's own shader TEXT -- the strings composeShaders()
 * (decoders/jsv.js:2459-2470) assembles from decoders/shaders/mpeg1video.js:18-29, and the
 * vertex / colour-conversion shaders of player/parts/end.js:77-166 -- mechanically, so that
 * golden vectors for dequantisation + IDCT can be produced without any hand transcription of
 * the shader arithmetic.  This file knows the LANGUAGE (GLSL ES 1.00 section 4-6, 8), not the
 * shaders: it contains no constant, formula or identifier of the reference.
 *
 * Supported subset (everything those shaders use, rejected loudly otherwise):
 *   types       float int bool vec2 vec3 vec4 mat4 sampler2D, one-dimensional arrays
 *   qualifiers  precision statements (ignored), uniform / varying / attribute / const globals
 *   statements  declarations with initialisers, expression statements, if / else, for,
 *               continue, break, return, blocks
 *   expressions literals, constructors, function calls, [] indexing of arrays / vectors,
 *               swizzles (rgba, xyzw, stpq) as r- and l-values, unary - ! ++ --, * / + -,
 *               relational, == !=, && ||, ?:, = += -= *= /=
 *   built-ins   texture2D floor mod sign abs min max dot sqrt cos sin clamp fract
 *
 * Arithmetic model (SURVEY.md 8c, decisions D1 and D2):
 *   float = IEEE-754 binary32, every operation individually rounded to nearest-even
 *           (Math.fround after each + - * / and each built-in step), no contraction;
 *           mod(x, y) = x - y * floor(x / y) as GLSL ES 1.00 section 8.3 defines it.
 *   int   = 32-bit two's complement; '/' truncates toward zero; int(float) truncates.
 * GLSL ES 1.00 has no implicit conversions: mixed int/float operands are a compile error here
 * too, which doubles as a check that the text is being read the way a GLSL compiler reads it.
 */
'use strict';

const fr = Math.fround;

// ------------------------------------------------------------------ tokenizer
function tokenize(src) {
  const toks = [];
  const re = /\s+|\/\/[^\n]*|\/\*[\s\S]*?\*\/|((?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?)|([A-Za-z_]\w*)|(\+\+|--|<=|>=|==|!=|&&|\|\||\+=|-=|\*=|\/=|[-+*\/<>=!?:;,.(){}\[\]])/y;
  let pos = 0;
  while (pos < src.length) {
    re.lastIndex = pos;
    const m = re.exec(src);
    if (!m) throw new Error('glsl: cannot tokenize at ' + JSON.stringify(src.slice(pos, pos + 30)));
    pos = re.lastIndex;
    if (m[1] !== undefined) {
      const isFloat = /[.eE]/.test(m[1]);
      toks.push({ k: isFloat ? 'float' : 'int', v: isFloat ? fr(parseFloat(m[1])) : parseInt(m[1], 10) | 0 });
    } else if (m[2] !== undefined) toks.push({ k: 'id', v: m[2] });
    else if (m[3] !== undefined) toks.push({ k: 'op', v: m[3] });
  }
  toks.push({ k: 'eof', v: '<eof>' });
  return toks;
}

// ------------------------------------------------------------------ types
const VEC_N = { vec2: 2, vec3: 3, vec4: 4 };
const isVec = (t) => t in VEC_N;
const BASIC = new Set(['float', 'int', 'bool', 'vec2', 'vec3', 'vec4', 'mat4', 'sampler2D', 'void']);
const typeName = (t) => (typeof t === 'string' ? t : t.base + '[' + t.n + ']');
const sameType = (a, b) => typeName(a) === typeName(b);

function zeroOf(t) {
  if (typeof t !== 'string') { const a = new Array(t.n); for (let i = 0; i < t.n; i++) a[i] = zeroOf(t.base); return a; }
  if (t === 'float' || t === 'int') return 0;
  if (t === 'bool') return false;
  if (isVec(t)) return new Array(VEC_N[t]).fill(0);
  if (t === 'mat4') return new Array(16).fill(0);
  if (t === 'sampler2D') return 0;      // unit 0 until uniform1i says otherwise [2.10.4]
  return null;
}
function copyVal(t, v) {
  if (typeof t !== 'string') return v.map((e) => copyVal(t.base, e));
  if (isVec(t) || t === 'mat4') return v.slice();
  return v;
}

// ------------------------------------------------------------------ parser -> closures
const CONT = 1, BRK = 2, RET = 3;

class Compiler {
  constructor(src, samplerFetch) {
    this.t = tokenize(src);
    this.p = 0;
    this.globals = new Map();       // name -> {type, slot, qual}
    this.gcount = 0;
    this.ginit = [];                // closures (G) run once per program instance
    this.funcs = new Map();         // name -> {ret, params, nslots, body}
    this.scopes = null;
    this.fetch = samplerFetch;      // (samplerValue, u, v) -> [r,g,b,a] floats
    this.declareGlobal('gl_FragColor', 'vec4', 'out');
    this.declareGlobal('gl_Position', 'vec4', 'out');
  }
  err(msg) { const tk = this.t[this.p]; throw new Error('glsl: ' + msg + ' near token ' + this.p + ' (' + tk.v + ')'); }
  peek(v) { const tk = this.t[this.p]; return tk.k !== 'float' && tk.k !== 'int' && tk.v === v; }
  accept(v) { if (this.peek(v)) { this.p++; return true; } return false; }
  expect(v) { if (!this.accept(v)) this.err('expected ' + v); }
  ident() { const tk = this.t[this.p]; if (tk.k !== 'id') this.err('expected identifier'); this.p++; return tk.v; }
  isTypeTok() { const tk = this.t[this.p]; return tk.k === 'id' && BASIC.has(tk.v); }

  declareGlobal(name, type, qual) {
    const g = { type, slot: this.gcount++, qual };
    this.globals.set(name, g);
    return g;
  }
  declareLocal(name, type) {
    const s = { type, slot: this.fn.nslots++ };
    this.scopes[this.scopes.length - 1].set(name, s);
    return s;
  }
  lookup(name) {
    if (this.scopes) for (let i = this.scopes.length - 1; i >= 0; i--) if (this.scopes[i].has(name)) return { local: true, ...this.scopes[i].get(name) };
    if (this.globals.has(name)) return { local: false, ...this.globals.get(name) };
    this.err('undeclared identifier ' + name);
  }

  // ---- translation unit
  parseProgram() {
    while (this.t[this.p].k !== 'eof') {
      if (this.accept('precision')) { this.ident(); this.ident(); this.expect(';'); continue; }
      let qual = null;
      for (;;) {
        const tk = this.t[this.p];
        if (tk.k === 'id' && ['uniform', 'varying', 'attribute', 'const'].includes(tk.v)) { qual = tk.v; this.p++; }
        else if (tk.k === 'id' && ['highp', 'mediump', 'lowp'].includes(tk.v)) this.p++;
        else break;
      }
      if (!this.isTypeTok()) this.err('expected a type at global scope');
      const type = this.ident();
      const name = this.ident();
      if (this.peek('(')) { this.parseFunction(type, name); continue; }
      this.p--;       // re-read the first declarator name
      this.parseDeclarators(type, true, qual);
    }
    return this;
  }

  parseDeclarators(base, global, qual) {
    const stmts = [];
    do {
      const name = this.ident();
      let type = base;
      if (this.accept('[')) {
        const tk = this.t[this.p++];
        if (tk.k !== 'int') this.err('array size must be an integer literal');
        this.expect(']');
        type = { base, n: tk.v };
      }
      let init = null;
      if (this.accept('=')) {
        init = this.parseAssign();
        if (!sameType(init.type, type)) this.err('initialiser of ' + name + ' is ' + typeName(init.type) + ', not ' + typeName(type));
      }
      if (global) {
        const g = this.declareGlobal(name, type, qual);
        const slot = g.slot, z = type;
        if (init) { const f = init.get; this.ginit.push((G) => { G[slot] = copyVal(z, f(null, G)); }); }
        else this.ginit.push((G) => { G[slot] = zeroOf(z); });
      } else {
        const s = this.declareLocal(name, type);
        const slot = s.slot, z = type;
        if (init) { const f = init.get; stmts.push((L, G) => { L[slot] = copyVal(z, f(L, G)); return 0; }); }
        else stmts.push((L) => { L[slot] = zeroOf(z); return 0; });
      }
    } while (this.accept(','));
    this.expect(';');
    return stmts;
  }

  parseFunction(ret, name) {
    const fn = { ret, params: [], nslots: 0, body: null, name };
    this.fn = fn;
    this.scopes = [new Map()];
    this.expect('(');
    if (!this.peek(')')) {
      if (this.peek('void')) this.p++;
      else do {
        while (['in', 'const', 'highp', 'mediump', 'lowp'].includes(this.t[this.p].v)) this.p++;
        const pt = this.ident();
        if (!BASIC.has(pt)) this.err('bad parameter type ' + pt);
        const pn = this.ident();
        let type = pt;
        if (this.accept('[')) { const tk = this.t[this.p++]; this.expect(']'); type = { base: pt, n: tk.v }; }
        const s = this.declareLocal(pn, type);
        fn.params.push({ type, slot: s.slot });
      } while (this.accept(','));
    }
    this.expect(')');
    this.funcs.set(name, fn);      // before the body: direct recursion is illegal in GLSL anyway
    fn.body = this.parseBlock();
    this.scopes = null;
    this.fn = null;
  }

  // ---- statements: closures (L, G) -> completion code
  parseBlock() {
    this.expect('{');
    this.scopes.push(new Map());
    const list = [];
    while (!this.accept('}')) list.push(...this.parseStatement());
    this.scopes.pop();
    return seq(list);
  }

  parseStatement() {
    if (this.peek('{')) return [this.parseBlock()];
    if (this.accept(';')) return [];
    if (this.accept('if')) {
      this.expect('(');
      const c = this.parseExpr();
      if (c.type !== 'bool') this.err('if condition is ' + typeName(c.type));
      this.expect(')');
      const a = seq(this.parseStatement());
      let b = null;
      if (this.accept('else')) b = seq(this.parseStatement());
      const cf = c.get;
      return [b ? (L, G) => (cf(L, G) ? a(L, G) : b(L, G)) : (L, G) => (cf(L, G) ? a(L, G) : 0)];
    }
    if (this.accept('for')) {
      this.expect('(');
      this.scopes.push(new Map());
      let init = [];
      if (this.isTypeTok()) { const ty = this.ident(); init = this.parseDeclarators(ty, false, null); }
      else if (!this.accept(';')) { const e = this.parseExpr(); this.expect(';'); init = [exprStmt(e)]; }
      const cond = this.peek(';') ? null : this.parseExpr();
      this.expect(';');
      const step = this.peek(')') ? null : this.parseExpr();
      this.expect(')');
      const body = seq(this.parseStatement());
      this.scopes.pop();
      const i0 = seq(init), cf = cond ? cond.get : () => true, sf = step ? step.get : () => 0;
      return [(L, G) => {
        i0(L, G);
        for (; cf(L, G); sf(L, G)) {
          const r = body(L, G);
          if (r === BRK) break;
          if (r === RET) return RET;
        }
        return 0;
      }];
    }
    if (this.accept('continue')) { this.expect(';'); return [() => CONT]; }
    if (this.accept('break')) { this.expect(';'); return [() => BRK]; }
    if (this.accept('return')) {
      if (this.accept(';')) return [() => RET];
      const e = this.parseExpr();
      this.expect(';');
      if (!sameType(e.type, this.fn.ret)) this.err('return type mismatch in ' + this.fn.name);
      const f = e.get, ty = e.type;
      return [(L, G) => { L.ret = copyVal(ty, f(L, G)); return RET; }];
    }
    // a precision qualifier may precede the type of a local declaration [4.5.2]; it changes nothing here
    while (this.t[this.p].k === 'id' && ['highp', 'mediump', 'lowp'].includes(this.t[this.p].v) && this.t[this.p + 1].k === 'id' && BASIC.has(this.t[this.p + 1].v)) this.p++;
    if (this.isTypeTok() && this.t[this.p + 1].k === 'id') { const ty = this.ident(); return this.parseDeclarators(ty, false, null); }
    if (this.peek('const')) { this.p++; while (['highp', 'mediump', 'lowp'].includes(this.t[this.p].v)) this.p++; const ty = this.ident(); return this.parseDeclarators(ty, false, 'const'); }
    const e = this.parseExpr();
    this.expect(';');
    return [exprStmt(e)];
  }

  // ---- expressions: {type, get(L,G), set?(L,G,v)}
  parseExpr() { return this.parseAssign(); }

  parseAssign() {
    const lhs = this.parseTernary();
    for (const op of ['=', '+=', '-=', '*=', '/=']) {
      if (this.accept(op)) {
        if (!lhs.set) this.err('left side of ' + op + ' is not an l-value');
        const rhs = this.parseAssign();
        let val = rhs;
        if (op !== '=') val = this.binary(op[0], lhs, rhs);
        if (!sameType(val.type, lhs.type)) this.err('cannot assign ' + typeName(val.type) + ' to ' + typeName(lhs.type));
        const vf = val.get, sf = lhs.set, ty = lhs.type;
        return { type: ty, get: (L, G) => { const v = copyVal(ty, vf(L, G)); sf(L, G, v); return v; } };
      }
    }
    return lhs;
  }

  parseTernary() {
    const c = this.parseBin(0);
    if (!this.accept('?')) return c;
    if (c.type !== 'bool') this.err('?: condition is ' + typeName(c.type));
    const a = this.parseAssign();
    this.expect(':');
    const b = this.parseAssign();
    if (!sameType(a.type, b.type)) this.err('?: arms differ: ' + typeName(a.type) + ' / ' + typeName(b.type));
    const cf = c.get, af = a.get, bf = b.get;
    return { type: a.type, get: (L, G) => (cf(L, G) ? af(L, G) : bf(L, G)) };
  }

  parseBin(level) {
    const LEVELS = [['||'], ['&&'], ['==', '!='], ['<', '>', '<=', '>='], ['+', '-'], ['*', '/']];
    if (level === LEVELS.length) return this.parseUnary();
    let lhs = this.parseBin(level + 1);
    for (;;) {
      const op = LEVELS[level].find((o) => this.peek(o));
      if (!op) return lhs;
      this.p++;
      const rhs = this.parseBin(level + 1);
      lhs = this.binary(op, lhs, rhs);
    }
  }

  binary(op, a, b) {
    const af = a.get, bf = b.get;
    if (op === '||' || op === '&&') {
      if (a.type !== 'bool' || b.type !== 'bool') this.err(op + ' needs bool operands');
      return { type: 'bool', get: op === '||' ? (L, G) => af(L, G) || bf(L, G) : (L, G) => af(L, G) && bf(L, G) };
    }
    if (op === '==' || op === '!=') {
      if (!sameType(a.type, b.type)) this.err(op + ' on ' + typeName(a.type) + ' and ' + typeName(b.type));
      const eq = isVec(a.type) ? (x, y) => x.every((e, i) => e === y[i]) : (x, y) => x === y;
      return { type: 'bool', get: op === '==' ? (L, G) => eq(af(L, G), bf(L, G)) : (L, G) => !eq(af(L, G), bf(L, G)) };
    }
    if (['<', '>', '<=', '>='].includes(op)) {
      if (a.type !== b.type || (a.type !== 'float' && a.type !== 'int')) this.err(op + ' on ' + typeName(a.type) + ' and ' + typeName(b.type));
      switch (op) {
        case '<': return { type: 'bool', get: (L, G) => af(L, G) < bf(L, G) };
        case '>': return { type: 'bool', get: (L, G) => af(L, G) > bf(L, G) };
        case '<=': return { type: 'bool', get: (L, G) => af(L, G) <= bf(L, G) };
        default: return { type: 'bool', get: (L, G) => af(L, G) >= bf(L, G) };
      }
    }
    // arithmetic
    if (a.type === 'int' && b.type === 'int') {
      switch (op) {
        case '+': return { type: 'int', get: (L, G) => (af(L, G) + bf(L, G)) | 0 };
        case '-': return { type: 'int', get: (L, G) => (af(L, G) - bf(L, G)) | 0 };
        case '*': return { type: 'int', get: (L, G) => Math.imul(af(L, G), bf(L, G)) };
        default: return { type: 'int', get: (L, G) => { const d = bf(L, G); if (d === 0) throw new Error('glsl: integer division by zero'); return Math.trunc(af(L, G) / d) | 0; } };
      }
    }
    const fop = { '+': (x, y) => fr(x + y), '-': (x, y) => fr(x - y), '*': (x, y) => fr(x * y), '/': (x, y) => fr(x / y) }[op];
    if (a.type === 'float' && b.type === 'float') return { type: 'float', get: (L, G) => fop(af(L, G), bf(L, G)) };
    if (isVec(a.type) && a.type === b.type) return { type: a.type, get: (L, G) => { const x = af(L, G), y = bf(L, G); return x.map((e, i) => fop(e, y[i])); } };
    if (isVec(a.type) && b.type === 'float') return { type: a.type, get: (L, G) => { const x = af(L, G), y = bf(L, G); return x.map((e) => fop(e, y)); } };
    if (a.type === 'float' && isVec(b.type)) return { type: b.type, get: (L, G) => { const x = af(L, G), y = bf(L, G); return y.map((e) => fop(x, e)); } };
    if (op === '*' && a.type === 'vec4' && b.type === 'mat4') {
      // row vector times matrix (GLSL ES 1.00 5.11): component j = dot(v, column j), summed left to right
      return { type: 'vec4', get: (L, G) => {
        const v = af(L, G), m = bf(L, G), o = [0, 0, 0, 0];
        for (let j = 0; j < 4; j++) { let s = fr(v[0] * m[4 * j]); for (let i = 1; i < 4; i++) s = fr(s + fr(v[i] * m[4 * j + i])); o[j] = s; }
        return o;
      } };
    }
    this.err('operator ' + op + ' on ' + typeName(a.type) + ' and ' + typeName(b.type));
  }

  parseUnary() {
    if (this.accept('-')) {
      const e = this.parseUnary(), f = e.get;
      if (e.type === 'int') return { type: 'int', get: (L, G) => (-f(L, G)) | 0 };
      if (e.type === 'float') return { type: 'float', get: (L, G) => -f(L, G) };
      if (isVec(e.type)) return { type: e.type, get: (L, G) => f(L, G).map((x) => -x) };
      this.err('unary - on ' + typeName(e.type));
    }
    if (this.accept('+')) return this.parseUnary();
    if (this.accept('!')) {
      const e = this.parseUnary(), f = e.get;
      if (e.type !== 'bool') this.err('! on ' + typeName(e.type));
      return { type: 'bool', get: (L, G) => !f(L, G) };
    }
    if (this.peek('++') || this.peek('--')) {
      const d = this.t[this.p++].v === '++' ? 1 : -1;
      const e = this.parseUnary();
      return this.incdec(e, d, true);
    }
    return this.parsePostfix();
  }

  incdec(e, d, pre) {
    if (!e.set || (e.type !== 'int' && e.type !== 'float')) this.err('++/-- needs a scalar l-value');
    const g = e.get, s = e.set, isInt = e.type === 'int';
    return { type: e.type, get: (L, G) => { const old = g(L, G); const nv = isInt ? (old + d) | 0 : fr(old + d); s(L, G, nv); return pre ? nv : old; } };
  }

  parsePostfix() {
    let e = this.parsePrimary();
    for (;;) {
      if (this.accept('[')) {
        const ix = this.parseExpr();
        this.expect(']');
        if (ix.type !== 'int') this.err('index is ' + typeName(ix.type));
        e = this.indexed(e, ix);
      } else if (this.accept('.')) {
        e = this.swizzle(e, this.ident());
      } else if (this.peek('++') || this.peek('--')) {
        const d = this.t[this.p++].v === '++' ? 1 : -1;
        e = this.incdec(e, d, false);
      } else return e;
    }
  }

  indexed(e, ix) {
    let et, n;
    if (typeof e.type !== 'string') { et = e.type.base; n = e.type.n; }
    else if (isVec(e.type)) { et = 'float'; n = VEC_N[e.type]; }
    else this.err('cannot index ' + typeName(e.type));
    const g = e.get, xf = ix.get;
    const chk = (i) => { if (i < 0 || i >= n) throw new Error('glsl: index ' + i + ' out of range 0..' + (n - 1)); return i; };
    const out = { type: et, get: (L, G) => g(L, G)[chk(xf(L, G))] };
    // containers are held by reference in their slot, so writing through get() updates the variable
    if (e.set || e.ref) { out.set = (L, G, v) => { g(L, G)[chk(xf(L, G))] = v; }; out.ref = true; }
    return out;
  }

  swizzle(e, name) {
    if (!isVec(e.type)) this.err('swizzle .' + name + ' on ' + typeName(e.type));
    const n = VEC_N[e.type];
    const sets = ['rgba', 'xyzw', 'stpq'];
    const set = sets.find((s) => [...name].every((ch) => s.includes(ch)));
    if (!set || name.length > 4) this.err('bad swizzle .' + name);
    const idx = [...name].map((ch) => set.indexOf(ch));
    if (idx.some((i) => i >= n)) this.err('swizzle .' + name + ' exceeds ' + e.type);
    const g = e.get;
    if (idx.length === 1) {
      const i0 = idx[0];
      const out = { type: 'float', get: (L, G) => g(L, G)[i0] };
      if (e.set || e.ref) { out.set = (L, G, v) => { g(L, G)[i0] = v; }; }
      return out;
    }
    const ty = 'vec' + idx.length;
    const out = { type: ty, get: (L, G) => { const v = g(L, G); return idx.map((i) => v[i]); } };
    if ((e.set || e.ref) && new Set(idx).size === idx.length) out.set = (L, G, val) => { const v = g(L, G); idx.forEach((i, k) => { v[i] = val[k]; }); };
    return out;
  }

  parsePrimary() {
    const tk = this.t[this.p];
    if (tk.k === 'float') { this.p++; const v = tk.v; return { type: 'float', get: () => v }; }
    if (tk.k === 'int') { this.p++; const v = tk.v; return { type: 'int', get: () => v }; }
    if (this.accept('(')) { const e = this.parseExpr(); this.expect(')'); return e; }
    if (tk.k !== 'id') this.err('unexpected token');
    this.p++;
    const name = tk.v;
    if (name === 'true' || name === 'false') { const v = name === 'true'; return { type: 'bool', get: () => v }; }
    if (this.accept('(')) {
      const args = [];
      if (!this.peek(')')) do args.push(this.parseAssign()); while (this.accept(','));
      this.expect(')');
      return this.call(name, args);
    }
    const s = this.lookup(name);
    const slot = s.slot, ty = s.type;
    const writable = s.local || s.qual === null || s.qual === 'out' || s.qual === 'varying' || s.qual === undefined;
    const e = s.local ? { type: ty, get: (L) => L[slot] } : { type: ty, get: (L, G) => G[slot] };
    if (writable) e.set = s.local ? (L, G, v) => { L[slot] = v; } : (L, G, v) => { G[slot] = v; };
    return e;
  }

  call(name, args) {
    const at = args.map((a) => typeName(a.type));
    const f = args.map((a) => a.get);
    const bad = () => this.err('no overload ' + name + '(' + at.join(', ') + ')');
    // ---- constructors
    if (name === 'float' || name === 'int' || name === 'bool') {
      if (args.length !== 1 || !['float', 'int', 'bool'].includes(at[0])) bad();
      const g = f[0], from = at[0];
      if (name === 'float') return { type: 'float', get: from === 'bool' ? (L, G) => (g(L, G) ? 1 : 0) : from === 'int' ? (L, G) => fr(g(L, G)) : g };
      if (name === 'int') return { type: 'int', get: from === 'float' ? (L, G) => { const x = g(L, G); if (!(Math.abs(x) < 2147483648)) throw new Error('glsl: int() of ' + x); return Math.trunc(x) | 0; } : from === 'bool' ? (L, G) => (g(L, G) ? 1 : 0) : g };
      return { type: 'bool', get: (L, G) => g(L, G) !== 0 && g(L, G) !== false };
    }
    if (isVec(name) || name === 'mat4') {
      const n = name === 'mat4' ? 16 : VEC_N[name];
      const flat = (L, G) => {
        const o = [];
        for (let i = 0; i < args.length; i++) {
          const v = f[i](L, G), t = at[i];
          if (t === 'float') o.push(v); else if (t === 'int') o.push(fr(v)); else if (t === 'bool') o.push(v ? 1 : 0); else if (isVec(t)) o.push(...v); else bad();
        }
        return o;
      };
      const count = at.reduce((s, t) => s + (isVec(t) ? VEC_N[t] : 1), 0);
      if (args.length === 1 && !isVec(at[0]) && name !== 'mat4') return { type: name, get: (L, G) => new Array(n).fill(flat(L, G)[0]) };
      if (count < n) bad();
      return { type: name, get: (L, G) => flat(L, G).slice(0, n) };
    }
    // ---- built-ins (GLSL ES 1.00 section 8), component-wise on vectors
    const gentype = (fn1) => {
      if (args.length !== 1) bad();
      if (at[0] === 'float') return { type: 'float', get: (L, G) => fn1(f[0](L, G)) };
      if (isVec(at[0])) return { type: at[0], get: (L, G) => f[0](L, G).map(fn1) };
      bad();
    };
    const gentype2 = (fn2) => {
      if (args.length !== 2) bad();
      if (at[0] === 'float' && at[1] === 'float') return { type: 'float', get: (L, G) => fn2(f[0](L, G), f[1](L, G)) };
      if (isVec(at[0]) && at[1] === at[0]) return { type: at[0], get: (L, G) => { const y = f[1](L, G); return f[0](L, G).map((x, i) => fn2(x, y[i])); } };
      if (isVec(at[0]) && at[1] === 'float') return { type: at[0], get: (L, G) => { const y = f[1](L, G); return f[0](L, G).map((x) => fn2(x, y)); } };
      bad();
    };
    switch (name) {
      case 'floor': return gentype((x) => Math.floor(x));
      case 'abs': return gentype((x) => Math.abs(x));
      case 'sign': return gentype((x) => (x > 0 ? 1 : x < 0 ? -1 : 0));
      case 'fract': return gentype((x) => fr(x - Math.floor(x)));
      case 'sqrt': return gentype((x) => fr(Math.sqrt(x)));
      case 'cos': return gentype((x) => fr(Math.cos(x)));
      case 'sin': return gentype((x) => fr(Math.sin(x)));
      case 'mod': return gentype2((x, y) => fr(x - fr(y * Math.floor(fr(x / y)))));
      case 'min': return gentype2((x, y) => (y < x ? y : x));
      case 'max': return gentype2((x, y) => (x < y ? y : x));
      case 'clamp':
        if (args.length !== 3 || at[0] !== 'float' || at[1] !== 'float' || at[2] !== 'float') bad();
        return { type: 'float', get: (L, G) => Math.min(Math.max(f[0](L, G), f[1](L, G)), f[2](L, G)) };
      case 'dot':
        if (args.length !== 2 || at[0] !== at[1] || !(isVec(at[0]) || at[0] === 'float')) bad();
        if (at[0] === 'float') return { type: 'float', get: (L, G) => fr(f[0](L, G) * f[1](L, G)) };
        return { type: 'float', get: (L, G) => { const x = f[0](L, G), y = f[1](L, G); let s = fr(x[0] * y[0]); for (let i = 1; i < x.length; i++) s = fr(s + fr(x[i] * y[i])); return s; } };
      case 'texture2D': {
        if (args.length !== 2 || at[0] !== 'sampler2D' || at[1] !== 'vec2') bad();
        const fetch = this.fetch;
        return { type: 'vec4', get: (L, G) => { const c = f[1](L, G); return fetch(f[0](L, G), c[0], c[1]); } };
      }
      default:
    }
    // ---- user functions (by value in, by value out)
    const fn = this.funcs.get(name);
    if (!fn) this.err('unknown function ' + name);
    if (fn.params.length !== args.length) bad();
    fn.params.forEach((p, i) => { if (!sameType(p.type, args[i].type)) bad(); });
    const params = fn.params;
    return { type: fn.ret, get: (L, G) => {
      const NL = new Array(fn.nslots);
      for (let i = 0; i < params.length; i++) NL[params[i].slot] = copyVal(params[i].type, f[i](L, G));
      fn.body(NL, G);
      return NL.ret;
    } };
  }
}

function seq(list) {
  if (list.length === 1) return list[0];
  const n = list.length;
  return (L, G) => {
    for (let i = 0; i < n; i++) { const r = list[i](L, G); if (r) return r; }
    return 0;
  };
}
function exprStmt(e) { const f = e.get; return (L, G) => { f(L, G); return 0; }; }

/*
 * compile(source, fetch) -> program object:
 *   .globals            Map name -> {type, slot, qual}
 *   .instantiate()      -> G (global storage, initialisers run)
 *   .main(G)            runs void main()
 * `fetch(samplerValue, u, v)` implements texture2D; sampler uniforms hold whatever the caller
 * stores in their slot (softgl.js stores the texture-unit number).
 */
function compile(source, fetch) {
  const c = new Compiler(source, fetch).parseProgram();
  const main = c.funcs.get('main');
  if (!main) throw new Error('glsl: no main()');
  return {
    globals: c.globals,
    instantiate() { const G = new Array(c.gcount); for (const f of c.ginit) f(G); G[c.globals.get('gl_FragColor').slot] = [0, 0, 0, 0]; G[c.globals.get('gl_Position').slot] = [0, 0, 0, 1]; return G; },
    main(G) { const L = new Array(main.nslots); main.body(L, G); },
  };
}

module.exports = { compile, tokenize };
