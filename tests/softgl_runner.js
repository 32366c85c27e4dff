#!/usr/bin/env node
/*
 * softgl_runner.js -- runs the conformance cases of tests/test_softgl.py on tools/softgl/{glsl,softgl}.js.
 * Test tooling.  Reads a JSON list of cases from the file named on the command line, prints a JSON list of results.
 *   {kind: 'glsl', src}                      compile src (a whole fragment shader) with glsl.compile, run main(), return
 *                                            gl_FragColor as four numbers (binary32 values, printed exactly as doubles)
 *   {kind: 'gl', textures, fs, w, h, unpack} draw the 4-vertex strip over a w x h RGBA8 colour attachment with the given
 *                                            fragment shader; textures: [{unit, w, h, fmt, bytes, filter}]; the vertex shader
 *                                            hands the varying `uv` = position * 0.5 + 0.5.  Returns the attachment's bytes.
 * An exception becomes {error: message}.
 */
'use strict';
const fs = require('fs');
const path = require('path');
const glsl = require(path.join(__dirname, '..', 'tools', 'softgl', 'glsl.js'));
const softgl = require(path.join(__dirname, '..', 'tools', 'softgl', 'softgl.js'));

const VS = 'attribute vec2 pos; varying vec2 uv; void main() { uv = pos * 0.5 + vec2(0.5, 0.5); gl_Position = vec4(pos, 0.0, 1.0); }';

function runGl(c) {
  const gl = softgl.createContext({ width: c.w, height: c.h });
  const sh = (type, src) => { const s = gl.createShader(type); gl.shaderSource(s, src); gl.compileShader(s); if (!gl.getShaderParameter(s, gl.COMPILE_STATUS)) throw new Error(gl.getShaderInfoLog(s)); return s; };
  const p = gl.createProgram();
  gl.attachShader(p, sh(gl.VERTEX_SHADER, c.vs || VS));
  gl.attachShader(p, sh(gl.FRAGMENT_SHADER, c.fs));
  gl.linkProgram(p);
  if (!gl.getProgramParameter(p, gl.LINK_STATUS)) throw new Error('link: ' + gl.getProgramInfoLog(p));
  gl.useProgram(p);
  const buf = gl.createBuffer();
  gl.bindBuffer(gl.ARRAY_BUFFER, buf);
  gl.bufferData(gl.ARRAY_BUFFER, new Float32Array([-1, -1, -1, 1, 1, -1, 1, 1]), gl.STATIC_DRAW);
  const loc = gl.getAttribLocation(p, 'pos');
  gl.enableVertexAttribArray(loc);
  gl.vertexAttribPointer(loc, 2, gl.FLOAT, false, 0, 0);
  if (c.unpack) gl.pixelStorei(gl.UNPACK_ALIGNMENT, c.unpack);
  for (const t of c.textures || []) {
    const tex = gl.createTexture();
    gl.activeTexture(gl.TEXTURE0 + t.unit);
    gl.bindTexture(gl.TEXTURE_2D, tex);
    const fmt = { rgba: gl.RGBA, l: gl.LUMINANCE, la: gl.LUMINANCE_ALPHA }[t.fmt];
    if (t.filter !== 'default') {
      gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MAG_FILTER, gl.NEAREST);
      gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MIN_FILTER, gl.NEAREST);
      gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_WRAP_S, gl.CLAMP_TO_EDGE);
      gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_WRAP_T, gl.CLAMP_TO_EDGE);
    }
    if (t.bytes) gl.texImage2D(gl.TEXTURE_2D, 0, fmt, t.w, t.h, 0, fmt, gl.UNSIGNED_BYTE, new Uint8Array(t.bytes));
    gl.uniform1i(gl.getUniformLocation(p, t.name || ('t' + t.unit)), t.unit);
  }
  for (const [name, v] of Object.entries(c.uniforms1f || {})) gl.uniform1f(gl.getUniformLocation(p, name), v);
  const target = gl.createTexture();
  gl.activeTexture(gl.TEXTURE0 + 15);
  gl.bindTexture(gl.TEXTURE_2D, target);
  gl.texImage2D(gl.TEXTURE_2D, 0, gl.RGBA, c.w, c.h, 0, gl.RGBA, gl.UNSIGNED_BYTE, null);
  gl.bindTexture(gl.TEXTURE_2D, null);
  const fbo = gl.createFramebuffer();
  gl.bindFramebuffer(gl.FRAMEBUFFER, fbo);
  gl.framebufferTexture2D(gl.FRAMEBUFFER, gl.COLOR_ATTACHMENT0, gl.TEXTURE_2D, target, 0);
  gl.viewport(0, 0, c.w, c.h);
  gl.drawArrays(gl.TRIANGLE_STRIP, 0, 4);
  return { bytes: Array.from(gl._textureBytes(target)) };
}

function runGlsl(c) {
  const prog = glsl.compile(c.src, () => [0, 0, 0, 1]);
  const G = prog.instantiate();
  prog.main(G);
  return { color: G[prog.globals.get('gl_FragColor').slot].map(Number) };
}

const cases = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const out = cases.map((c) => {
  try { return c.kind === 'gl' ? runGl(c) : runGlsl(c); } catch (e) { return { error: String(e.message) }; }
});
process.stdout.write(JSON.stringify(out));
