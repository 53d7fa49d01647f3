"""Stand-in for the slice of MLX that the reference's model code touches, backed by PyTorch on the CPU -- FIXTURE GENERATION ONLY
(tests/golden/make_reference_mlx_fixtures.py).  MLX itself (Apple-only, `mlx==0.17.3`, setup.py:32 of the reference) cannot be
installed in the build container; with this module registered as `mlx`, `mlx.core`, `mlx.nn` and `mlx.utils`, the reference's
own python/src/diffusionkit/mlx/{config,mmdit,sampler,vae}.py import and run unmodified, so its WIRING (block structure, modulation
order, RoPE tables, QK-norm placement, joint-sequence order, patchify / unpatchify, schedules) is executed rather than restated.
What this cannot reproduce is MLX's arithmetic: everything here is plain float32 torch (dtype arguments are honoured by casting),
so fixtures are generated with float32 configs and compared against the oracle's exact-math mode.

Each operation follows the documented MLX semantics of the same name (numpy-style broadcasting / repeat / split, `transpose` =
permutation of all axes, channels-last convolutions, `mx.fast.layer_norm(x, weight, bias, eps)` over the last axis,
`mx.fast.scaled_dot_product_attention(q, k, v, scale=...)` on [B, H, S, D], `nn.RMSNorm`: x * rsqrt(mean(x^2) + eps) * weight,
`nn.GELU()` = exact erf form, `nn.GroupNorm(..., pytorch_compatible=True)` = torch's grouping).
"""
import math
import sys
import types

import numpy as np
import torch


class Dtype:
    def __init__(self, name, t, size):
        self.name, self.t, self.size = name, t, size

    def __repr__(self):
        return f"mlx.core.{self.name}"


float32 = Dtype("float32", torch.float32, 4)
float16 = Dtype("float16", torch.float16, 2)
bfloat16 = Dtype("bfloat16", torch.bfloat16, 2)
int32 = Dtype("int32", torch.int32, 4)
int64 = Dtype("int64", torch.int64, 8)
uint8 = Dtype("uint8", torch.uint8, 1)
bool_ = Dtype("bool", torch.bool, 1)
_BY_TORCH = {d.t: d for d in (float32, float16, bfloat16, int32, int64, uint8, bool_)}
_BY_TORCH[torch.float64] = float32


def _t(x):
    if isinstance(x, array):
        return x.t
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(x, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(x))
        return t.float() if t.dtype == torch.float64 else (t.int() if t.dtype == torch.int64 else t)
    if isinstance(x, (bool, int)) and not isinstance(x, float):
        return torch.tensor(x, dtype=torch.int32)
    if isinstance(x, float) or isinstance(x, np.floating):
        return torch.tensor(float(x), dtype=torch.float32)
    if isinstance(x, (list, tuple)):
        if len(x) == 0:
            return torch.zeros(0, dtype=torch.float32)
        return _t(np.asarray([np.asarray(_t(e)) if isinstance(e, (array, torch.Tensor)) else e for e in x]))
    raise TypeError(f"cannot make an array from {type(x)}")


def _scalar_like(other, ref):
    """python scalars take the array's dtype (MLX weak typing); ints stay ints only against int arrays"""
    if isinstance(other, (int, float, np.floating, np.integer)) and not isinstance(other, bool):
        if ref.dtype.is_floating_point or isinstance(other, (float, np.floating)):
            return torch.tensor(float(other), dtype=ref.dtype if ref.dtype.is_floating_point else torch.float32)
        return torch.tensor(int(other), dtype=ref.dtype)
    return _t(other)


class array:
    def __init__(self, data, dtype=None):
        t = _t(data)
        self.t = t.to(dtype.t) if dtype is not None else t

    # -- properties
    shape = property(lambda s: tuple(s.t.shape))
    ndim = property(lambda s: s.t.dim())
    size = property(lambda s: s.t.numel())
    dtype = property(lambda s: _BY_TORCH[s.t.dtype])

    def __len__(self):
        return self.t.shape[0]

    def __iter__(self):
        for i in range(self.t.shape[0]):
            yield array(self.t[i])

    @staticmethod
    def _index(idx):
        if isinstance(idx, tuple):
            return tuple(_t(i).long() if isinstance(i, array) else i for i in idx)
        return idx.t.long() if isinstance(idx, array) else idx

    def __getitem__(self, idx):
        return array(self.t[self._index(idx)])

    def __setitem__(self, idx, value):
        self.t = self.t.clone()
        self.t[self._index(idx)] = _scalar_like(value, self.t).to(self.t.dtype)

    def item(self):
        return self.t.item()

    def tolist(self):
        return self.t.tolist()

    def __array__(self, dtype=None, copy=None):
        a = self.t.detach().float().numpy() if self.t.dtype in (torch.bfloat16, torch.float16) else self.t.detach().numpy()
        return a.astype(dtype) if dtype is not None else a

    # -- methods
    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return array(self.t.reshape(*shape))

    def transpose(self, *axes):
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        if not axes:
            axes = tuple(reversed(range(self.t.dim())))
        return array(self.t.permute(*axes))

    T = property(lambda s: s.transpose())

    def astype(self, dtype):
        return array(self.t.to(dtype.t))

    def squeeze(self, axis=None):
        return array(self.t.squeeze() if axis is None else self.t.squeeze(axis))

    def flatten(self, start_axis=0, end_axis=-1):
        return array(torch.flatten(self.t, start_axis, end_axis))

    def sum(self, axis=None, keepdims=False):
        return array(self.t.sum() if axis is None else self.t.sum(axis, keepdim=keepdims))

    def mean(self, axis=None, keepdims=False):
        return array(self.t.mean() if axis is None else self.t.mean(axis, keepdim=keepdims))

    def max(self, axis=None, keepdims=False):
        return array(self.t.max() if axis is None else self.t.amax(axis, keepdim=keepdims))

    def min(self, axis=None, keepdims=False):
        return array(self.t.min() if axis is None else self.t.amin(axis, keepdim=keepdims))

    def square(self):
        return array(self.t * self.t)

    def argmax(self, axis=None, keepdims=False):
        return array((self.t.argmax() if axis is None else self.t.argmax(dim=axis, keepdim=keepdims)).int())

    def split(self, indices_or_sections, axis=0):
        return split(self, indices_or_sections, axis)

    # -- arithmetic
    def _bin(self, other, fn, rev=False):
        o = _scalar_like(other, self.t)
        a, b = (o, self.t) if rev else (self.t, o)
        return array(fn(a, b))

    __add__ = lambda s, o: s._bin(o, torch.add)
    __radd__ = lambda s, o: s._bin(o, torch.add, True)
    __sub__ = lambda s, o: s._bin(o, torch.sub)
    __rsub__ = lambda s, o: s._bin(o, torch.sub, True)
    __mul__ = lambda s, o: s._bin(o, torch.mul)
    __rmul__ = lambda s, o: s._bin(o, torch.mul, True)
    __pow__ = lambda s, o: s._bin(o, torch.pow)
    __rpow__ = lambda s, o: s._bin(o, torch.pow, True)
    __matmul__ = lambda s, o: s._bin(o, torch.matmul)
    __neg__ = lambda s: array(-s.t)
    __lt__ = lambda s, o: s._bin(o, torch.lt)
    __le__ = lambda s, o: s._bin(o, torch.le)
    __gt__ = lambda s, o: s._bin(o, torch.gt)
    __ge__ = lambda s, o: s._bin(o, torch.ge)

    def __truediv__(self, o):
        a = self.t if self.t.dtype.is_floating_point else self.t.float()
        return array(a / _scalar_like(o, a))

    def __rtruediv__(self, o):
        a = self.t if self.t.dtype.is_floating_point else self.t.float()
        return array(_scalar_like(o, a) / a)

    def __floordiv__(self, o):
        return self._bin(o, lambda a, b: torch.div(a, b, rounding_mode="floor"))

    def __eq__(self, o):  # noqa: D105
        return self._bin(o, torch.eq)

    __hash__ = None

    def __repr__(self):
        return f"array({self.t})"


def _float(x):
    t = _t(x)
    return t if t.dtype.is_floating_point else t.float()


def arange(start, stop=None, step=1, dtype=None):
    if stop is None:
        start, stop = 0, start
    floaty = any(isinstance(v, float) for v in (start, stop, step))
    t = torch.arange(start, stop, step, dtype=torch.float32 if floaty else torch.int32)
    return array(t if dtype is None else t.to(dtype.t))


def linspace(start, stop, num=50, dtype=float32):
    return array(torch.linspace(float(start), float(stop), int(num), dtype=torch.float32).to(dtype.t))


def zeros(shape, dtype=float32):
    return array(torch.zeros(*((shape,) if isinstance(shape, int) else tuple(shape)), dtype=dtype.t))


def ones(shape, dtype=float32):
    return array(torch.ones(*((shape,) if isinstance(shape, int) else tuple(shape)), dtype=dtype.t))


def _promote(ts):
    dt = ts[0].dtype
    for t in ts[1:]:
        dt = torch.promote_types(dt, t.dtype)
    return [t.to(dt) for t in ts]


def concatenate(arrays, axis=0):
    return array(torch.cat(_promote([_t(a) for a in arrays]), dim=axis))


def stack(arrays, axis=0):
    return array(torch.stack(_promote([_t(a) for a in arrays]), dim=axis))


def split(a, indices_or_sections, axis=0):
    t = _t(a)
    if isinstance(indices_or_sections, int):
        assert t.shape[axis] % indices_or_sections == 0
        return [array(p) for p in torch.split(t, t.shape[axis] // indices_or_sections, dim=axis)]
    return [array(p) for p in torch.tensor_split(t, list(indices_or_sections), dim=axis)]


def repeat(a, repeats, axis=None):
    t = _t(a)
    return array(torch.repeat_interleave(t.flatten() if axis is None else t, repeats, dim=0 if axis is None else axis))


def expand_dims(a, axis):
    t = _t(a)
    for ax in sorted(axis) if isinstance(axis, (tuple, list)) else [axis]:
        t = t.unsqueeze(ax)
    return array(t)


def pad(a, pad_width, constant_values=0):
    t = _t(a)
    flat = []
    for lo, hi in reversed([tuple(p) for p in pad_width]):
        flat += [lo, hi]
    return array(torch.nn.functional.pad(t, flat, value=constant_values))


def broadcast_to(a, shape):
    return array(torch.broadcast_to(_t(a), tuple(shape)))


def minimum(a, b):
    return array(a)._bin(b, torch.minimum) if not isinstance(a, array) else a._bin(b, torch.minimum)


def maximum(a, b):
    return array(a)._bin(b, torch.maximum) if not isinstance(a, array) else a._bin(b, torch.maximum)


def where(c, a, b):
    ct = _t(c).bool()
    ref = _t(a) if isinstance(a, (array, torch.Tensor)) else (_t(b) if isinstance(b, (array, torch.Tensor)) else torch.zeros(()))
    return array(torch.where(ct, _scalar_like(a, ref), _scalar_like(b, ref)))


abs = lambda a: array(torch.abs(_t(a)))  # noqa: A001
zeros_like = lambda a: array(torch.zeros_like(_t(a)))
ones_like = lambda a: array(torch.ones_like(_t(a)))


def clip(a, lo, hi):
    return array(torch.clamp(_t(a), lo, hi))


def softmax(a, axis=-1, precise=False):
    t = _t(a)
    return array(torch.softmax(t.float(), dim=axis).to(t.dtype))


exp = lambda a: array(torch.exp(_float(a)))
log = lambda a: array(torch.log(_float(a)))
sin = lambda a: array(torch.sin(_float(a)))
cos = lambda a: array(torch.cos(_float(a)))
sqrt = lambda a: array(torch.sqrt(_float(a)))
rsqrt = lambda a: array(torch.rsqrt(_float(a)))
sigmoid = lambda a: array(torch.sigmoid(_float(a)))
erf = lambda a: array(torch.erf(_float(a)))
square = lambda a: array(_t(a) * _t(a))
einsum = lambda eq, *ops: array(torch.einsum(eq, *[_t(o) for o in ops]))
eval = lambda *a, **k: None  # noqa: A001  (mx.eval: nothing is lazy here)


def mean(a, axis=None, keepdims=False):
    return array(a).mean(axis, keepdims) if not isinstance(a, array) else a.mean(axis, keepdims)


class _Fast:
    @staticmethod
    def layer_norm(x, weight, bias, eps):
        t = _t(x)
        f = t.float()
        mu = f.mean(-1, keepdim=True)
        var = ((f - mu) ** 2).mean(-1, keepdim=True)
        y = (f - mu) * torch.rsqrt(var + eps)
        if weight is not None:
            y = y * _t(weight).float()
        if bias is not None:
            y = y + _t(bias).float()
        return array(y.to(t.dtype))

    @staticmethod
    def rms_norm(x, weight, eps):
        t = _t(x)
        f = t.float()
        y = f * torch.rsqrt((f * f).mean(-1, keepdim=True) + eps)
        if weight is not None:
            y = y * _t(weight).float()
        return array(y.to(t.dtype))

    @staticmethod
    def scaled_dot_product_attention(q, k, v, *, scale, mask=None, memory_efficient_threshold=None):
        qt, kt, vt = _t(q), _t(k), _t(v)
        s = (qt.float() * scale) @ kt.float().transpose(-1, -2)
        if mask is not None:
            s = s + _t(mask).float()
        return array((torch.softmax(s, dim=-1) @ vt.float()).to(qt.dtype))


fast = _Fast()


class _Random:
    @staticmethod
    def seed(s):
        torch.manual_seed(int(s))

    @staticmethod
    def normal(shape=(), dtype=float32, loc=0.0, scale=1.0, key=None):
        return array((torch.randn(*tuple(shape)) * scale + loc).to(dtype.t))


random = _Random()


class _Metal:  # memory bookkeeping of the Apple GPU backend: nothing to report here
    device_info = staticmethod(lambda: {"memory_size": 0, "max_recommended_working_set_size": 0})
    set_memory_limit = staticmethod(lambda *a, **k: 0)
    set_cache_limit = staticmethod(lambda *a, **k: 0)
    get_peak_memory = staticmethod(lambda: 0)
    get_active_memory = staticmethod(lambda: 0)
    get_cache_memory = staticmethod(lambda: 0)
    reset_peak_memory = staticmethod(lambda: None)
    clear_cache = staticmethod(lambda: None)


metal = _Metal()
int16 = Dtype("int16", torch.int16, 2)
uint32 = Dtype("uint32", torch.int64, 4)
_BY_TORCH[torch.int16] = int16


def load(*a, **k):
    raise RuntimeError("mx.load: no checkpoints in the fixture generator")


# ---- mlx.utils ------------------------------------------------------------------------------------------------------------
def tree_map(fn, tree, *rest):
    if isinstance(tree, dict):
        return {k: tree_map(fn, v, *[r[k] for r in rest]) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(tree_map(fn, v, *[r[i] for r in rest]) for i, v in enumerate(tree))
    return fn(tree, *rest)


def tree_unflatten(items):
    """[('a.0.b', v), ...] -> nested dicts / lists (numeric keys become list positions)"""
    root = {}
    for name, v in items:
        parts = name.split(".")
        cur = root
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = v

    def fix(node):
        if not isinstance(node, dict):
            return node
        if node and all(k.isdigit() for k in node):
            return [fix(node[str(i)]) if str(i) in node else {} for i in range(max(int(k) for k in node) + 1)]
        return {k: fix(v) for k, v in node.items()}

    return fix(root)


def tree_flatten(tree, prefix=""):
    out = []
    if isinstance(tree, dict):
        for k, v in tree.items():
            out += tree_flatten(v, f"{prefix}.{k}" if prefix else str(k))
    elif isinstance(tree, (list, tuple)):
        for i, v in enumerate(tree):
            out += tree_flatten(v, f"{prefix}.{i}" if prefix else str(i))
    else:
        out.append((prefix, tree))
    return out


# ---- mlx.nn ---------------------------------------------------------------------------------------------------------------
class Module:
    """Attribute-based module: arrays are parameters, Modules / lists of Modules are children (as mlx.nn.Module)."""

    def __init__(self):
        pass

    def __call__(self, *a, **k):
        raise NotImplementedError

    def _items(self):
        return [(k, v) for k, v in vars(self).items() if not k.startswith("_")]

    def __contains__(self, key):  # mlx.nn.Module is a dict of its attributes: `"upsample" in self`
        return key in vars(self)

    def parameters(self):
        def walk(v):
            if isinstance(v, array):
                return v
            if isinstance(v, Module):
                return v.parameters()
            if isinstance(v, (list, tuple)) and any(isinstance(e, (Module, array, list, tuple)) for e in v):
                return [walk(e) for e in v]
            return None

        out = {}
        for k, v in self._items():
            w = walk(v)
            if w is not None and not (isinstance(w, (dict, list)) and len(w) == 0):
                out[k] = w
        return out

    def update(self, params):
        def apply(holder, key, val):
            if not isinstance(holder, list) and key not in vars(holder):
                return  # mlx.nn.Module.update only touches what the module already has (a k_proj.bias for a bias-free Linear is dropped)
            cur = holder[key] if isinstance(holder, list) else getattr(holder, key)
            if isinstance(val, array):
                if isinstance(holder, list):
                    holder[key] = val
                else:
                    setattr(holder, key, val)
            elif isinstance(val, dict):
                cur.update(val)
            elif isinstance(val, (list, tuple)):
                for i, e in enumerate(val):
                    apply(cur, i, e)

        for k, v in params.items():
            apply(self, k, v)
        return self

    def named_modules(self):
        out = [("", self)]

        def walk(prefix, v):
            if isinstance(v, Module):
                for n, m in v.named_modules():
                    out.append((f"{prefix}.{n}" if n else prefix, m))
            elif isinstance(v, (list, tuple)):
                for i, e in enumerate(v):
                    walk(f"{prefix}.{i}", e)

        for k, v in self._items():
            walk(k, v)
        return out

    def modules(self):
        return [m for _, m in self.named_modules()]

    def children(self):
        return {k: v for k, v in self._items() if isinstance(v, (Module, list))}

    def eval(self):
        return self

    def load_weights(self, weights, strict=True):
        items = list(weights.items()) if isinstance(weights, dict) else list(weights)
        if strict:
            have = {k for k, _ in tree_flatten(self.parameters())}
            assert have == {k for k, _ in items}, sorted(have ^ {k for k, _ in items})[:5]
        self.update(tree_unflatten(items))
        return self

    def set_dtype(self, dtype):
        self.update(tree_map(lambda a: a.astype(dtype) if a.t.dtype.is_floating_point else a, self.parameters()))


def _init(*shape):
    return array(torch.zeros(*shape))


class Linear(Module):
    def __init__(self, input_dims, output_dims, bias=True):
        super().__init__()
        self.weight = _init(output_dims, input_dims)
        if bias:
            self.bias = _init(output_dims)

    def __call__(self, x):
        y = _t(x) @ self.weight.t.transpose(0, 1)
        if "bias" in vars(self):
            y = y + self.bias.t
        return array(y)


class Embedding(Module):
    def __init__(self, num_embeddings, dims):
        super().__init__()
        self.weight = _init(num_embeddings, dims)

    def __call__(self, x):
        return array(self.weight.t[_t(x).long()])


class Conv2d(Module):
    """channels-last: input [B, H, W, C], weight [O, kh, kw, I]"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__()
        k = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.weight = _init(out_channels, k[0], k[1], in_channels)
        if bias:
            self.bias = _init(out_channels)
        self.stride = (stride, stride) if isinstance(stride, int) else tuple(stride)
        self.padding = (padding, padding) if isinstance(padding, int) else tuple(padding)
        self.dilation = dilation

    def __call__(self, x):
        t = _t(x).permute(0, 3, 1, 2)
        w = self.weight.t.permute(0, 3, 1, 2)
        b = self.bias.t if "bias" in vars(self) else None
        y = torch.nn.functional.conv2d(t, w, b, stride=self.stride, padding=self.padding, dilation=self.dilation)
        return array(y.permute(0, 2, 3, 1))


class GroupNorm(Module):
    def __init__(self, num_groups, dims, eps=1e-5, affine=True, pytorch_compatible=False):
        super().__init__()
        assert pytorch_compatible, "only the PyTorch grouping is modelled (what the reference asks for)"
        self.num_groups, self.dims, self.eps = num_groups, dims, eps
        if affine:
            self.weight = array(torch.ones(dims))
            self.bias = _init(dims)

    def __call__(self, x):
        t = _t(x)
        y = torch.nn.functional.group_norm(t.movedim(-1, 1).float(), self.num_groups, None, None, self.eps).movedim(1, -1)
        if "weight" in vars(self):
            y = y * self.weight.t + self.bias.t
        return array(y.to(t.dtype))


class LayerNorm(Module):
    def __init__(self, dims, eps=1e-5, affine=True, bias=True):
        super().__init__()
        self.eps = eps
        if affine:
            self.weight = array(torch.ones(dims))
            if bias:
                self.bias = _init(dims)

    def __call__(self, x):
        return fast.layer_norm(x, vars(self).get("weight"), vars(self).get("bias"), self.eps)


class RMSNorm(Module):
    def __init__(self, dims, eps=1e-5):
        super().__init__()
        self.weight = array(torch.ones(dims))
        self.eps = eps

    def __call__(self, x):
        return fast.rms_norm(x, self.weight, self.eps)


class SiLU(Module):
    def __call__(self, x):
        t = _t(x)
        return array(t * torch.sigmoid(t))


class GELU(Module):
    def __init__(self, approx="none"):
        super().__init__()
        assert approx == "none"

    def __call__(self, x):
        t = _t(x)
        return array(0.5 * t * (1.0 + torch.erf(t / math.sqrt(2.0))))


class Identity(Module):
    def __init__(self, *a, **k):
        super().__init__()

    def __call__(self, x, *a, **k):
        return x


class Sequential(Module):
    def __init__(self, *modules):
        super().__init__()
        self.layers = list(modules)

    def __call__(self, x):
        for m in self.layers:
            x = m(x)
        return x


class MultiHeadAttention(Module):
    """mlx.nn.MultiHeadAttention: projections without bias unless asked, heads split from the last axis, additive mask."""

    def __init__(self, dims, num_heads, query_input_dims=None, key_input_dims=None, value_input_dims=None, value_dims=None,
                 value_output_dims=None, bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.query_proj = Linear(query_input_dims or dims, dims, bias=bias)
        self.key_proj = Linear(key_input_dims or dims, dims, bias=bias)
        self.value_proj = Linear(value_input_dims or key_input_dims or dims, value_dims or dims, bias=bias)
        self.out_proj = Linear(value_dims or dims, value_output_dims or dims, bias=bias)

    def __call__(self, queries, keys, values, mask=None):
        q, k, v = self.query_proj(queries), self.key_proj(keys), self.value_proj(values)
        H = self.num_heads
        B, L, _ = q.shape
        S = k.shape[1]
        q = q.reshape(B, L, H, -1).transpose(0, 2, 1, 3)
        k = k.reshape(B, S, H, -1).transpose(0, 2, 1, 3)
        v = v.reshape(B, S, H, -1).transpose(0, 2, 1, 3)
        o = fast.scaled_dot_product_attention(q, k, v, scale=math.sqrt(1.0 / q.shape[-1]), mask=mask)
        return self.out_proj(o.transpose(0, 2, 1, 3).reshape(B, L, -1))


def silu(x):
    return SiLU()(x)


def relu(x):
    return array(torch.relu(_t(x)))


def gelu_fast_approx(x):
    t = _t(x)
    return array(t * torch.sigmoid(1.702 * t))


def quantize(*a, **k):
    raise RuntimeError("nn.quantize: quantised checkpoints are outside the fixture generator")


def gelu(x):
    return GELU()(x)


def install():
    """Registers this module's contents as mlx / mlx.core / mlx.nn / mlx.utils."""
    me = sys.modules[__name__]
    core = types.ModuleType("mlx.core")
    for n in ("array", "Dtype", "float32", "float16", "bfloat16", "int32", "int64", "uint8", "bool_", "arange", "zeros", "ones",
              "concatenate", "stack", "split", "repeat", "expand_dims", "pad", "broadcast_to", "clip", "softmax", "exp", "log", "sin", "cos", "sqrt",
              "rsqrt", "sigmoid", "erf", "square", "einsum", "eval", "mean", "fast", "linspace", "random", "minimum", "maximum", "where",
              "abs", "zeros_like", "ones_like", "metal", "int16", "uint32", "load"):
        setattr(core, n, getattr(me, n))
    core.float = float32  # (the reference only ever writes mx.float32 / mx.float16; kept for attribute scans)
    nn = types.ModuleType("mlx.nn")
    for n in ("Module", "Linear", "Embedding", "Conv2d", "GroupNorm", "LayerNorm", "RMSNorm", "SiLU", "GELU", "Identity", "Sequential",
              "silu", "gelu", "relu", "gelu_fast_approx", "MultiHeadAttention", "quantize"):
        setattr(nn, n, getattr(me, n))
    utils = types.ModuleType("mlx.utils")
    utils.tree_map, utils.tree_flatten, utils.tree_unflatten = tree_map, tree_flatten, tree_unflatten
    mlx = types.ModuleType("mlx")
    mlx.core, mlx.nn, mlx.utils = core, nn, utils
    for m in (mlx, core, nn, utils):
        sys.modules[m.__name__] = m
    return core, nn
