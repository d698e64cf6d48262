"""Stand-in for the `moderngl` package (test infrastructure, build container only).

The reference's renderer (/root/reference/rgbd_3d/moderngl_renderer.py) drives OpenGL through moderngl on an EGL
device; neither exists in this image, but Mesa's software rasteriser does.  This module implements exactly the moderngl
API subset that file uses -- on top of an off-screen llvmpipe context (oracle/glshim/glctx.c) and raw OpenGL calls
through ctypes -- so that tests/golden/make_golden_gl.py can import and run the reference renderer UNCHANGED, with its
own GLSL shaders, and store what real OpenGL rasterisation produces as golden vectors.

Semantics follow moderngl 5.x: texture formats (components x 'f4' -> GL_R*32F, depth_texture -> DEPTH_COMPONENT24),
NEAREST/clamp parameters, `vertex_array` attribute formats ('3f 3f 2f 1f' interleaved), `Uniform.value` / `.write`,
`Texture.use` / `bind_to_image` / `read` / `write`, `Framebuffer.use` / `clear`, `ComputeShader.run`.  Deliberate
differences: depth textures keep GL_TEXTURE_COMPARE_MODE = NONE (the reference fetches them through a plain sampler2D
with texelFetch: comparison mode would make that undefined), and image-store -> fetch hazards between dispatches get the
glMemoryBarrier the specification requires (moderngl issues none; the synchronous software rasteriser does not care).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(os.path.dirname(_HERE), "_build", "libglctx.so")
_DRIVER = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so"

# moderngl's enable flags / constants (values as in moderngl)
NOTHING, BLEND, DEPTH_TEST, CULL_FACE = 0, 1, 2, 4
NEAREST, LINEAR = 0x2600, 0x2601
TRIANGLES = 0x0004

_GL = dict(
    DEPTH_TEST=0x0B71, CULL_FACE=0x0B44, BLEND=0x0BE2, LESS=0x0201, LEQUAL=0x0203, CCW=0x0901, CW=0x0900,
    COLOR_BUFFER_BIT=0x4000, DEPTH_BUFFER_BIT=0x100, ARRAY_BUFFER=0x8892, ELEMENT_ARRAY_BUFFER=0x8893, DYNAMIC_DRAW=0x88E8,
    FLOAT=0x1406, UNSIGNED_INT=0x1405, VERTEX_SHADER=0x8B31, FRAGMENT_SHADER=0x8B30, COMPUTE_SHADER=0x91B9,
    COMPILE_STATUS=0x8B81, LINK_STATUS=0x8B82, TEXTURE_2D=0x0DE1, TEXTURE0=0x84C0, TEXTURE_MIN_FILTER=0x2801,
    TEXTURE_MAG_FILTER=0x2800, TEXTURE_WRAP_S=0x2802, TEXTURE_WRAP_T=0x2803, CLAMP_TO_EDGE=0x812F, REPEAT=0x2901,
    RGBA32F=0x8814, RGB32F=0x8815, RG32F=0x8230, R32F=0x822E, RGBA=0x1908, RGB=0x1907, RG=0x8227, RED=0x1903,
    DEPTH_COMPONENT24=0x81A6, DEPTH_COMPONENT=0x1902, FRAMEBUFFER=0x8D40, COLOR_ATTACHMENT0=0x8CE0, DEPTH_ATTACHMENT=0x8D00,
    FRAMEBUFFER_COMPLETE=0x8CD5, READ_ONLY=0x88B8, WRITE_ONLY=0x88B9, READ_WRITE=0x88BA, UNPACK_ALIGNMENT=0x0CF5,
    PACK_ALIGNMENT=0x0D05, ALL_BARRIER_BITS=0xFFFFFFFF, ACTIVE_UNIFORMS=0x8B86, FLOAT_MAT4=0x8B5C, FLOAT_VEC3=0x8B51,
    VERSION=0x1F02, RENDERER=0x1F01)

_u, _i, _f, _vp, _cp = C.c_uint, C.c_int, C.c_float, C.c_void_p, C.c_char_p
_PROTO = {
    "glGetError": (_u, []), "glGetString": (_cp, [_u]), "glEnable": (None, [_u]), "glDisable": (None, [_u]),
    "glDepthFunc": (None, [_u]), "glFrontFace": (None, [_u]), "glViewport": (None, [_i, _i, _i, _i]),
    "glClearColor": (None, [_f, _f, _f, _f]), "glClearDepth": (None, [C.c_double]), "glClear": (None, [_u]),
    "glGenBuffers": (None, [_i, C.POINTER(_u)]), "glBindBuffer": (None, [_u, _u]),
    "glBufferData": (None, [_u, C.c_ssize_t, _vp, _u]), "glBufferSubData": (None, [_u, C.c_ssize_t, C.c_ssize_t, _vp]),
    "glGenVertexArrays": (None, [_i, C.POINTER(_u)]), "glBindVertexArray": (None, [_u]),
    "glEnableVertexAttribArray": (None, [_u]), "glVertexAttribPointer": (None, [_u, _i, _u, C.c_ubyte, _i, _vp]),
    "glCreateShader": (_u, [_u]), "glShaderSource": (None, [_u, _i, C.POINTER(_cp), C.POINTER(_i)]),
    "glCompileShader": (None, [_u]), "glGetShaderiv": (None, [_u, _u, C.POINTER(_i)]),
    "glGetShaderInfoLog": (None, [_u, _i, C.POINTER(_i), _cp]), "glCreateProgram": (_u, []), "glAttachShader": (None, [_u, _u]),
    "glLinkProgram": (None, [_u]), "glGetProgramiv": (None, [_u, _u, C.POINTER(_i)]),
    "glGetProgramInfoLog": (None, [_u, _i, C.POINTER(_i), _cp]), "glUseProgram": (None, [_u]),
    "glGetUniformLocation": (_i, [_u, _cp]), "glGetAttribLocation": (_i, [_u, _cp]), "glUniform1i": (None, [_i, _i]),
    "glUniform3fv": (None, [_i, _i, _vp]), "glUniformMatrix4fv": (None, [_i, _i, C.c_ubyte, _vp]),
    "glGetActiveUniform": (None, [_u, _u, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_u), _cp]),
    "glGenTextures": (None, [_i, C.POINTER(_u)]), "glBindTexture": (None, [_u, _u]), "glActiveTexture": (None, [_u]),
    "glTexImage2D": (None, [_u, _i, _i, _i, _i, _i, _u, _u, _vp]),
    "glTexSubImage2D": (None, [_u, _i, _i, _i, _i, _i, _u, _u, _vp]), "glTexParameteri": (None, [_u, _u, _i]),
    "glGetTexImage": (None, [_u, _i, _u, _u, _vp]), "glPixelStorei": (None, [_u, _i]),
    "glBindImageTexture": (None, [_u, _u, _i, C.c_ubyte, _i, _u, _u]),
    "glGenFramebuffers": (None, [_i, C.POINTER(_u)]), "glBindFramebuffer": (None, [_u, _u]),
    "glFramebufferTexture2D": (None, [_u, _u, _u, _u, _i]), "glCheckFramebufferStatus": (_u, [_u]),
    "glDrawElements": (None, [_u, _i, _u, _vp]), "glDispatchCompute": (None, [_u, _u, _u]),
    "glMemoryBarrier": (None, [_u]), "glFinish": (None, []),
}


class _Api:
    def __init__(self):
        if not os.path.exists(_SO):
            raise RuntimeError(f"{_SO} missing: `make -C oracle` builds it where Mesa's swrast driver and DRI header exist")
        lib = C.CDLL(_SO)
        lib.glctx_create.argtypes = [_cp, _i, _i, _i]
        lib.glctx_error.restype = _cp
        lib.glctx_proc.restype = _vp
        lib.glctx_proc.argtypes = [_cp]
        # compatibility profile: the reference's vertex / fragment shaders are `#version 130`
        if lib.glctx_create(_DRIVER.encode(), 4, 5, 1) != 0:
            raise RuntimeError("no off-screen OpenGL context: " + lib.glctx_error().decode())
        for name, (res, args) in _PROTO.items():
            p = lib.glctx_proc(name.encode())
            if not p:
                raise RuntimeError("OpenGL entry point missing: " + name)
            setattr(self, name[2:], C.CFUNCTYPE(res, *args)(p))
        self.version = self.GetString(_GL["VERSION"]).decode()
        self.renderer = self.GetString(_GL["RENDERER"]).decode()

    def check(self, what):
        e = self.GetError()
        if e:
            raise RuntimeError(f"OpenGL error 0x{e:04x} after {what}")


_api = None


def _gl():
    global _api
    if _api is None:
        _api = _Api()
    return _api


def _raw(data):
    """bytes of a buffer-like / numpy / glm stand-in object"""
    if hasattr(data, "to_gl_bytes"):
        return data.to_gl_bytes()
    if isinstance(data, (bytes, bytearray)):
        return bytes(data)
    return np.ascontiguousarray(data).tobytes()


def _compile(kind, src):
    g = _gl()
    sh = g.CreateShader(kind)
    s = src.encode()
    arr = (_cp * 1)(s)
    ln = (_i * 1)(len(s))
    g.ShaderSource(sh, 1, arr, ln)
    g.CompileShader(sh)
    ok = _i(0)
    g.GetShaderiv(sh, _GL["COMPILE_STATUS"], C.byref(ok))
    if not ok.value:
        log = C.create_string_buffer(8192)
        g.GetShaderInfoLog(sh, 8192, None, log)
        raise RuntimeError("GLSL compile error:\n" + log.value.decode())
    return sh


def _link(shaders):
    g = _gl()
    prog = g.CreateProgram()
    for sh in shaders:
        g.AttachShader(prog, sh)
    g.LinkProgram(prog)
    ok = _i(0)
    g.GetProgramiv(prog, _GL["LINK_STATUS"], C.byref(ok))
    if not ok.value:
        log = C.create_string_buffer(8192)
        g.GetProgramInfoLog(prog, 8192, None, log)
        raise RuntimeError("GLSL link error:\n" + log.value.decode())
    return prog


class Uniform:
    def __init__(self, prog, name, gltype):
        self.prog, self.name, self.gltype = prog, name, gltype
        self.loc = _gl().GetUniformLocation(prog, name.encode())

    @property
    def value(self):
        raise NotImplementedError

    @value.setter
    def value(self, v):
        g = _gl()
        g.UseProgram(self.prog)
        g.Uniform1i(self.loc, int(v))            # the reference only assigns sampler units
        g.check("Uniform.value")

    def write(self, data):
        g = _gl()
        raw = _raw(data)
        g.UseProgram(self.prog)
        buf = C.create_string_buffer(raw, len(raw))
        if self.gltype == _GL["FLOAT_MAT4"]:
            assert len(raw) == 64, (self.name, len(raw))
            g.UniformMatrix4fv(self.loc, 1, 0, buf)   # column-major bytes, no transpose (what moderngl does)
        elif self.gltype == _GL["FLOAT_VEC3"]:
            assert len(raw) == 12, (self.name, len(raw))
            g.Uniform3fv(self.loc, 1, buf)
        else:
            raise NotImplementedError(f"uniform {self.name}: type 0x{self.gltype:x}")
        g.check("Uniform.write " + self.name)


class _ProgramBase:
    def _collect(self):
        g = _gl()
        n = _i(0)
        g.GetProgramiv(self.glo, _GL["ACTIVE_UNIFORMS"], C.byref(n))
        self._uniforms = {}
        for k in range(n.value):
            nm = C.create_string_buffer(256)
            sz, ty, ln = _i(0), _u(0), _i(0)
            g.GetActiveUniform(self.glo, k, 256, C.byref(ln), C.byref(sz), C.byref(ty), nm)
            self._uniforms[nm.value.decode()] = Uniform(self.glo, nm.value.decode(), ty.value)

    def __getitem__(self, name):
        return self._uniforms[name]       # KeyError for an inactive uniform, like moderngl

    def release(self):
        pass


class Program(_ProgramBase):
    def __init__(self, vertex_shader, fragment_shader):
        self.glo = _link([_compile(_GL["VERTEX_SHADER"], vertex_shader), _compile(_GL["FRAGMENT_SHADER"], fragment_shader)])
        self._collect()


class ComputeShader(_ProgramBase):
    def __init__(self, source):
        self.glo = _link([_compile(_GL["COMPUTE_SHADER"], source)])
        self._collect()

    def run(self, group_x=1, group_y=1, group_z=1):
        g = _gl()
        g.UseProgram(self.glo)
        g.MemoryBarrier(_GL["ALL_BARRIER_BITS"])
        g.DispatchCompute(group_x, group_y, group_z)
        g.MemoryBarrier(_GL["ALL_BARRIER_BITS"])
        g.check("ComputeShader.run")


class Buffer:
    def __init__(self, reserve, dynamic):
        g = _gl()
        b = _u(0)
        g.GenBuffers(1, C.byref(b))
        self.glo, self.size = b.value, int(reserve)
        g.BindBuffer(_GL["ARRAY_BUFFER"], self.glo)
        g.BufferData(_GL["ARRAY_BUFFER"], self.size, None, _GL["DYNAMIC_DRAW"])
        g.check("Buffer")

    def write(self, data, offset=0):
        g = _gl()
        raw = _raw(data)
        assert offset + len(raw) <= self.size, "buffer overflow (moderngl raises here too)"
        g.BindBuffer(_GL["ARRAY_BUFFER"], self.glo)
        g.BufferSubData(_GL["ARRAY_BUFFER"], offset, len(raw), C.create_string_buffer(raw, len(raw)))
        g.check("Buffer.write")

    def release(self):
        pass


class VertexArray:
    def __init__(self, program, content, index_buffer):
        g = _gl()
        v = _u(0)
        g.GenVertexArrays(1, C.byref(v))
        self.glo, self.program = v.value, program
        g.BindVertexArray(self.glo)
        for buf, fmt, *names in content:
            counts = [int(tok[:-1]) for tok in fmt.split()]
            assert all(tok.endswith("f") for tok in fmt.split()) and len(counts) == len(names)
            stride = 4 * sum(counts)
            g.BindBuffer(_GL["ARRAY_BUFFER"], buf.glo)
            off = 0
            for cnt, name in zip(counts, names):
                loc = g.GetAttribLocation(program.glo, name.encode())
                if loc < 0:
                    raise KeyError(name)          # moderngl refuses attributes the program does not have
                g.EnableVertexAttribArray(loc)
                g.VertexAttribPointer(loc, cnt, _GL["FLOAT"], 0, stride, C.c_void_p(off))
                off += 4 * cnt
        g.BindBuffer(_GL["ELEMENT_ARRAY_BUFFER"], index_buffer.glo)
        g.BindVertexArray(0)
        g.check("VertexArray")

    def render(self, mode=TRIANGLES, vertices=-1):
        g = _gl()
        g.UseProgram(self.program.glo)
        g.BindVertexArray(self.glo)
        g.DrawElements(mode, int(vertices), _GL["UNSIGNED_INT"], None)
        g.BindVertexArray(0)
        g.check("VertexArray.render")

    def release(self):
        pass


_FMT = {1: ("R32F", "RED"), 2: ("RG32F", "RG"), 3: ("RGB32F", "RGB"), 4: ("RGBA32F", "RGBA")}


class Texture:
    def __init__(self, size, components, dtype="f4", depth=False):
        assert dtype == "f4"
        g = _gl()
        t = _u(0)
        g.GenTextures(1, C.byref(t))
        self.glo, self.size, self.components, self.depth = t.value, tuple(size), components, depth
        self.ctx = None
        if depth:
            self.internal, self.base = _GL["DEPTH_COMPONENT24"], _GL["DEPTH_COMPONENT"]
        else:
            self.internal, self.base = _GL[_FMT[components][0]], _GL[_FMT[components][1]]
        g.ActiveTexture(_GL["TEXTURE0"] + 15)    # a scratch unit: object set-up must not disturb the units in use
        g.BindTexture(_GL["TEXTURE_2D"], self.glo)
        g.TexImage2D(_GL["TEXTURE_2D"], 0, self.internal, self.size[0], self.size[1], 0, self.base, _GL["FLOAT"], None)
        # moderngl defaults: LINEAR / LINEAR, repeat -- the reference overrides both where it samples
        for p, v in ((_GL["TEXTURE_MIN_FILTER"], LINEAR), (_GL["TEXTURE_MAG_FILTER"], LINEAR)):
            g.TexParameteri(_GL["TEXTURE_2D"], p, v)
        g.check("Texture")

    def _bind_scratch(self):
        g = _gl()
        g.ActiveTexture(_GL["TEXTURE0"] + 15)
        g.BindTexture(_GL["TEXTURE_2D"], self.glo)

    @property
    def filter(self):
        raise NotImplementedError

    @filter.setter
    def filter(self, mm):
        self._bind_scratch()
        _gl().TexParameteri(_GL["TEXTURE_2D"], _GL["TEXTURE_MIN_FILTER"], mm[0])
        _gl().TexParameteri(_GL["TEXTURE_2D"], _GL["TEXTURE_MAG_FILTER"], mm[1])

    def _wrap(self, axis, repeat):
        self._bind_scratch()
        _gl().TexParameteri(_GL["TEXTURE_2D"], axis, _GL["REPEAT"] if repeat else _GL["CLAMP_TO_EDGE"])

    repeat_x = property(fset=lambda self, v: self._wrap(_GL["TEXTURE_WRAP_S"], v))
    repeat_y = property(fset=lambda self, v: self._wrap(_GL["TEXTURE_WRAP_T"], v))

    def write(self, data):
        g = _gl()
        raw = _raw(data)
        n = self.size[0] * self.size[1] * (1 if self.depth else self.components) * 4
        assert len(raw) == n, (len(raw), n)
        self._bind_scratch()
        g.PixelStorei(_GL["UNPACK_ALIGNMENT"], 1)
        g.TexSubImage2D(_GL["TEXTURE_2D"], 0, 0, 0, self.size[0], self.size[1], self.base, _GL["FLOAT"],
                        C.create_string_buffer(raw, len(raw)))
        g.check("Texture.write")

    def read(self):
        g = _gl()
        n = self.size[0] * self.size[1] * (1 if self.depth else self.components) * 4
        buf = C.create_string_buffer(n)
        g.MemoryBarrier(_GL["ALL_BARRIER_BITS"])
        g.Finish()
        self._bind_scratch()
        g.PixelStorei(_GL["PACK_ALIGNMENT"], 1)
        g.GetTexImage(_GL["TEXTURE_2D"], 0, self.base, _GL["FLOAT"], buf)
        g.check("Texture.read")
        return buf.raw

    def use(self, location=0):
        g = _gl()
        g.ActiveTexture(_GL["TEXTURE0"] + location)
        g.BindTexture(_GL["TEXTURE_2D"], self.glo)

    def bind_to_image(self, unit, read=True, write=True):
        acc = _GL["READ_WRITE"] if read and write else (_GL["READ_ONLY"] if read else _GL["WRITE_ONLY"])
        if self.ctx is not None:                 # image units are per-context state in moderngl (one GL context per renderer)
            self.ctx._image_bindings[unit] = (self, acc)
        _gl().BindImageTexture(unit, self.glo, 0, 0, 0, acc, self.internal)
        _gl().check("bind_to_image")

    def release(self):
        pass


class Framebuffer:
    def __init__(self, color_attachments, depth_attachment):
        g = _gl()
        f = _u(0)
        g.GenFramebuffers(1, C.byref(f))
        self.glo = f.value
        self.size = color_attachments[0].size
        self._viewport = (0, 0) + tuple(self.size)
        g.BindFramebuffer(_GL["FRAMEBUFFER"], self.glo)
        for k, t in enumerate(color_attachments):
            g.FramebufferTexture2D(_GL["FRAMEBUFFER"], _GL["COLOR_ATTACHMENT0"] + k, _GL["TEXTURE_2D"], t.glo, 0)
        if depth_attachment is not None:
            g.FramebufferTexture2D(_GL["FRAMEBUFFER"], _GL["DEPTH_ATTACHMENT"], _GL["TEXTURE_2D"], depth_attachment.glo, 0)
        st = g.CheckFramebufferStatus(_GL["FRAMEBUFFER"])
        if st != _GL["FRAMEBUFFER_COMPLETE"]:
            raise RuntimeError(f"framebuffer incomplete: 0x{st:x}")
        g.check("Framebuffer")

    def use(self):
        g = _gl()
        g.BindFramebuffer(_GL["FRAMEBUFFER"], self.glo)
        g.Viewport(*self._viewport)
        Context.current_fbo = self

    @property
    def viewport(self):
        return self._viewport

    @viewport.setter
    def viewport(self, v):
        self._viewport = tuple(int(x) for x in v)
        if Context.current_fbo is self:
            _gl().Viewport(*self._viewport)

    def clear(self, red=0.0, green=0.0, blue=0.0, alpha=0.0, depth=1.0):
        g = _gl()
        g.BindFramebuffer(_GL["FRAMEBUFFER"], self.glo)
        g.ClearColor(red, green, blue, alpha)
        g.ClearDepth(depth)
        g.Clear(_GL["COLOR_BUFFER_BIT"] | _GL["DEPTH_BUFFER_BIT"])
        if Context.current_fbo is not None and Context.current_fbo is not self:
            g.BindFramebuffer(_GL["FRAMEBUFFER"], Context.current_fbo.glo)
        g.check("Framebuffer.clear")

    def release(self):
        pass


class Context:
    current_fbo = None

    def __init__(self):
        self.info = {"GL_VERSION": _gl().version, "GL_RENDERER": _gl().renderer}
        self._image_bindings = {}

    def __enter__(self):
        # every moderngl context of the reference is its own GL context; here they share one: restore this one's image units
        for unit, (tex, acc) in self._image_bindings.items():
            _gl().BindImageTexture(unit, tex.glo, 0, 0, 0, acc, tex.internal)
        return self

    def __exit__(self, *a):
        return False

    def program(self, vertex_shader, fragment_shader):
        return Program(vertex_shader, fragment_shader)

    def compute_shader(self, source):
        return ComputeShader(source)

    def buffer(self, data=None, reserve=0, dynamic=False):
        b = Buffer(reserve if data is None else len(_raw(data)), dynamic)
        if data is not None:
            b.write(data)
        return b

    def vertex_array(self, program, content, index_buffer=None):
        return VertexArray(program, content, index_buffer)

    def texture(self, size, components, data=None, dtype="f4"):
        t = Texture(size, components, dtype)
        t.ctx = self
        if data is not None:
            t.write(data)
        return t

    def depth_texture(self, size, data=None):
        t = Texture(size, 1, "f4", depth=True)
        t.ctx = self
        return t

    def framebuffer(self, color_attachments=(), depth_attachment=None):
        return Framebuffer(list(color_attachments), depth_attachment)

    def clear(self, red=0.0, green=0.0, blue=0.0, alpha=0.0, depth=1.0):
        assert Context.current_fbo is not None, "the reference always clears after Framebuffer.use()"
        Context.current_fbo.clear(red, green, blue, alpha, depth)

    def enable(self, flags):
        for bit, cap in ((BLEND, "BLEND"), (DEPTH_TEST, "DEPTH_TEST"), (CULL_FACE, "CULL_FACE")):
            if flags & bit:
                _gl().Enable(_GL[cap])

    def disable(self, flags):
        for bit, cap in ((BLEND, "BLEND"), (DEPTH_TEST, "DEPTH_TEST"), (CULL_FACE, "CULL_FACE")):
            if flags & bit:
                _gl().Disable(_GL[cap])

    depth_func = property(fset=lambda self, v: _gl().DepthFunc({"<": _GL["LESS"], "<=": _GL["LEQUAL"]}[v]))
    front_face = property(fset=lambda self, v: _gl().FrontFace({"ccw": _GL["CCW"], "cw": _GL["CW"]}[v]))

    def release(self):
        pass


def create_context(standalone=True, backend=None, device_index=0, require=None, **kw):
    return Context()
