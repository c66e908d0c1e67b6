"""Reading and writing policies in the two formats rl-tools writes per checkpoint (SURVEY.md §5 "checkpoint /
resume", §8(f) row 3): the C++ code export `checkpoint.h` and its HDF5 twin `checkpoint.h5` (read with
the dependency-free minimal reader in `hdf5_min.py`).

rl-tools writes every parameter as
    namespace <path> { ... alignas(float) const unsigned char memory[] = {b0, b1, ...}; ... }
(little-endian float32 bytes, row-major; `checkpoint.h:1,39,50,75,...`).  ``load_checkpoint_header``
parses such a file supplied by the user and returns the flat parameter vector this engine takes
(order: W0[16,22] b0[16] Wi[48,16] Wh[48,16] bi[48] bh[48] h0[16] W2[4,16] b2[4]) plus, when present,
the embedded known-answer example (input [T,B,22], output [T,B,4]) for ``Raptor.selftest``.
Only the Dense(22->16, ReLU) -> GRU(16) -> Dense(16->4) topology of the RAPTOR policy is accepted.
"""
import re

import numpy as np

_LAYOUT = [
    ("actor/layer_0/weights", 16 * 22),
    ("actor/layer_0/biases", 16),
    ("actor/layer_1/weights_input", 48 * 16),
    ("actor/layer_1/weights_hidden", 48 * 16),
    ("actor/layer_1/biases_input", 48),
    ("actor/layer_1/biases_hidden", 48),
    ("actor/layer_1/initial_hidden_state", 16),
    ("actor/layer_2/weights", 4 * 16),
    ("actor/layer_2/biases", 4),
]


def parse_blobs(text):
    """{namespace path: float32 array} for every byte-array blob of a checkpoint header."""
    blobs, stack = {}, []
    ns = re.compile(r"^\s*namespace\s+([\w:]+)\s*\{\s*$")
    for line in text.split("\n"):
        m = ns.match(line)
        if m:
            stack.append(m.group(1).replace("rl_tools::checkpoint::", "").replace("::", "/"))
        elif line.strip() == "}":
            if stack:
                stack.pop()
        elif "const unsigned char memory[]" in line:
            body = line[line.index("{") + 1: line.rindex("}")]
            raw = np.array([int(t) for t in body.split(",")], dtype=np.uint8)
            if raw.size % 4:
                raise ValueError("blob size is not a multiple of 4 bytes")
            blobs["/".join(s for s in stack if s != "parameters_memory")] = raw.view("<f4").astype(np.float32)
    return blobs


def load_checkpoint_header(path):
    """-> (weights float32 [2084], example or None) with example = (input [T,B,22], output [T,B,4])."""
    blobs = parse_blobs(open(path).read())
    parts = []
    for name, size in _LAYOUT:
        if name not in blobs:
            raise ValueError(f"{path}: parameter '{name}' not found (not a RAPTOR-topology actor export?)")
        if blobs[name].size != size:
            raise ValueError(f"{path}: '{name}' has {blobs[name].size} values, expected {size}")
        parts.append(blobs[name])
    weights = np.concatenate(parts).astype(np.float32)
    example = None
    if "example/input" in blobs and "example/output" in blobs:
        x, y = blobs["example/input"], blobs["example/output"]
        m = re.search(r"example::input.*?Shape<[^,]+,\s*(\d+),\s*(\d+),\s*(\d+)>", open(path).read(), re.S)
        if m and int(m.group(3)) == 22:
            T, B = int(m.group(1)), int(m.group(2))
            if x.size == T * B * 22 and y.size == T * B * 4:
                example = (x.reshape(T, B, 22), y.reshape(T, B, 4))
    return weights, example


def write_checkpoint_header(path, weights, example=None):
    """Inverse of ``load_checkpoint_header`` for this topology: the byte-array blobs in the same
    namespaces (without the rl-tools type aliases, which need the rl-tools headers to mean anything)."""
    w = np.ascontiguousarray(weights, "<f4")
    if w.size != sum(s for _, s in _LAYOUT):
        raise ValueError("expected 2084 parameters")

    def blob(ns_path, arr):
        parts = ns_path.split("/")
        head = "namespace rl_tools::checkpoint::" + parts[0] + " {\n"
        for p in parts[1:]:
            head += "namespace " + p + " {\n"
        body = "alignas(float) const unsigned char memory[] = {" + ", ".join(str(b) for b in arr.tobytes()) + "};\n"
        return head + "namespace parameters_memory {\n" + body + "}\n" + "}\n" * len(parts)

    out, off = ["// NOTE: little-endian float32 byte arrays, row-major (out, in)\n"], 0
    for name, size in _LAYOUT:
        out.append(blob(name, w[off:off + size]))
        off += size
    if example is not None:
        x, y = (np.ascontiguousarray(a, "<f4") for a in example)
        T, B = x.shape[0], x.shape[1]
        for nm, arr, d in (("input", x, 22), ("output", y, 4)):
            out.append(f"namespace rl_tools::checkpoint::example::{nm} {{\n"
                       "alignas(float) const unsigned char memory[] = {" + ", ".join(str(b) for b in arr.tobytes()) + "};\n"
                       f"using SHAPE = rl_tools::tensor::Shape<unsigned long, {T}, {B}, {d}>;\n}}\n")
    open(path, "w").write("".join(out))


_H5_LAYOUT = [
    ("actor/layers/0/weights/parameters", (16, 22)), ("actor/layers/0/biases/parameters", (1, 16)),
    ("actor/layers/1/weights_input/parameters", (48, 16)), ("actor/layers/1/weights_hidden/parameters", (48, 16)),
    ("actor/layers/1/biases_input/parameters", (48,)), ("actor/layers/1/biases_hidden/parameters", (48,)),
    ("actor/layers/1/initial_hidden_state/parameters", (16,)),
    ("actor/layers/2/weights/parameters", (4, 16)), ("actor/layers/2/biases/parameters", (1, 4)),
]


def load_checkpoint_h5(path):
    """-> (weights float32 [2084], example or None, meta str or None) from an rl-tools `checkpoint.h5`
    (groups `/actor/layers/{0,1,2}`, string attributes `type` / `activation_function`; `h5:/actor...` in
    SURVEY.md).  Only Dense(22->16, RELU) -> GRU(16) -> Dense(16->4, IDENTITY) is accepted."""
    from .hdf5_min import File
    root = File(path).root
    try:
        actor = root["actor"]
        kinds = [(actor[f"layers/{i}"].attrs.get("type"), actor[f"layers/{i}"].attrs.get("activation_function"))
                 for i in range(3)]
    except KeyError as e:
        raise ValueError(f"{path}: not an rl-tools actor checkpoint (missing {e})")
    if actor.attrs.get("type") != "sequential" or kinds != [("dense", "RELU"), ("gru", None), ("dense", "IDENTITY")]:
        raise ValueError(f"{path}: unsupported topology {actor.attrs.get('type')} {kinds}; expected "
                         "sequential Dense(RELU) -> GRU -> Dense(IDENTITY)")
    if "layers/3" in actor:
        raise ValueError(f"{path}: more than three layers")
    parts = []
    for name, shape in _H5_LAYOUT:
        ds = root[name]
        if tuple(ds.shape) != shape or ds.dtype != np.dtype("<f4"):
            raise ValueError(f"{path}: {name} has shape {ds.shape} {ds.dtype}, expected {shape} float32")
        parts.append(ds.numpy().ravel())
    weights = np.concatenate(parts).astype(np.float32)
    example = None
    if "example/input" in root and "example/output" in root:
        x, y = root["example/input"].numpy(), root["example/output"].numpy()
        if x.ndim == 3 and x.shape[2] == 22 and y.shape == x.shape[:2] + (4,):
            example = (x.astype(np.float32), y.astype(np.float32))
    return weights, example, actor.attrs.get("meta")


_H5_ATTRS = {     # string attributes rl-tools attaches (tests/golden/checkpoint.h5)
    "actor": {"type": "sequential"},
    "actor/layers/0": {"activation_function": "RELU", "type": "dense"},
    "actor/layers/1": {"type": "gru"},
    "actor/layers/2": {"activation_function": "IDENTITY", "type": "dense"},
}
_DEFAULT_META = ('{"environment": {"name": "l2f","observation": "Position.OrientationRotationMatrix.LinearVelocity.'
                 'AngularVelocityDelayed(0).ActionHistory(1)"}}')


def _shape_attrs(shape, matrix):
    if matrix:
        return {"type": "matrix", "rows": str(shape[0]), "cols": str(shape[1])}
    a = {"type": "tensor", "num_dims": str(len(shape))}
    a.update({f"dim_{i}": str(d) for i, d in enumerate(shape)})
    return a


def write_checkpoint_h5(path, weights, example=None, meta=_DEFAULT_META, checkpoint_name=None):
    """Inverse of ``load_checkpoint_h5``: the flat 2 084-parameter vector (and optionally a known-answer
    example) in the group / dataset / attribute layout rl-tools writes, so a policy handled by this engine
    round-trips with the reference tooling.  Written with the dependency-free
    writer of `hdf5_min.py`; libhdf5's h5dump / h5diff accept the result (tests/test_host_logic.py)."""
    from .hdf5_min import DatasetSpec, GroupSpec, write_file
    w = np.ascontiguousarray(weights, np.float32).ravel()
    if w.size != sum(int(np.prod(sh)) for _, sh in _H5_LAYOUT):
        raise ValueError("expected 2084 parameters")
    tree, off = {}, 0

    def put(path_, node):
        parts = path_.split("/")
        d = tree
        for p_ in parts[:-1]:
            d = d.setdefault(p_, {})
        d[parts[-1]] = node

    for name, shape in _H5_LAYOUT:
        size = int(np.prod(shape))
        matrix = "/layers/0/" in name or "/layers/2/" in name          # dense layers store matrices, the GRU tensors
        put(name, DatasetSpec(w[off:off + size].reshape(shape), _shape_attrs(shape, matrix)))
        off += size
    if example is not None:
        x, y = (np.ascontiguousarray(a, np.float32) for a in example)
        if x.ndim != 3 or x.shape[2] != 22 or y.shape != x.shape[:2] + (4,):
            raise ValueError("example must be (input [T,B,22], output [T,B,4])")
        put("example/input", DatasetSpec(x, _shape_attrs(x.shape, False)))
        put("example/output", DatasetSpec(y, _shape_attrs(y.shape, False)))

    def build(d, prefix):
        attrs = dict(_H5_ATTRS.get(prefix, {}))
        if prefix == "actor":
            if checkpoint_name is not None:
                attrs["checkpoint_name"] = checkpoint_name
            if meta is not None:
                attrs["meta"] = meta
        return GroupSpec({k: (build(v, f"{prefix}/{k}" if prefix else k) if isinstance(v, dict) else v)
                          for k, v in d.items()}, attrs)

    write_file(path, build(tree, ""))


# rl-tools also knows FAST_TANH (an approximation of tanh whose definition is not in the reference tree): a teacher trained with it
# would be relabelled with a different function if it were mapped onto the exact tanh, so it is refused by name.
_ACT_NAMES = {"RELU": "relu", "TANH": "tanh", "IDENTITY": "identity"}


def load_mlp_checkpoint_h5(path, group="actor"):
    """A `sequential` of `dense` layers in rl-tools' HDF5 layout - what `extract_checkpoints.sh` gathers for the teachers
    (README.md:211-216), the layout of `h5:/actor/layers/*` in the shipped student checkpoint: groups `<group>/layers/{i}` with string
    attributes `type` = "dense" and `activation_function`, datasets `weights/parameters` [out, in] and `biases/parameters` [1, out]
    or [out].  -> (layers: list of (W [out, in], b [out]) float32, activations: list of "relu" | "tanh" | "identity").
    Anything else (a recurrent layer, a missing dataset, shapes that do not chain) is refused with the reason."""
    from .hdf5_min import File
    root = File(path).root
    if group not in root:
        raise ValueError(f"{path}: no group /{group}")
    g = root[group]
    if g.attrs.get("type") != "sequential":
        raise ValueError(f"{path}: /{group} is '{g.attrs.get('type')}', expected a 'sequential' model")
    layers, acts, i = [], [], 0
    while f"layers/{i}" in g:
        lay = g[f"layers/{i}"]
        kind, fn = lay.attrs.get("type"), lay.attrs.get("activation_function")
        if kind != "dense":
            raise ValueError(f"{path}: /{group}/layers/{i} is a '{kind}' layer; a teacher is a stack of dense layers")
        if fn == "FAST_TANH":
            raise ValueError(f"{path}: /{group}/layers/{i} uses FAST_TANH, rl-tools' tanh approximation; its definition is not in the "
                             "reference tree and evaluating it as the exact tanh would relabel with another function - refused")
        if fn not in _ACT_NAMES:
            raise ValueError(f"{path}: /{group}/layers/{i} has activation '{fn}' (supported: {sorted(_ACT_NAMES)})")
        try:
            W, b = lay["weights/parameters"].numpy(), lay["biases/parameters"].numpy()
        except KeyError as e:
            raise ValueError(f"{path}: /{group}/layers/{i} lacks {e}")
        W, b = np.asarray(W, np.float32), np.asarray(b, np.float32).reshape(-1)
        if W.ndim != 2 or b.shape[0] != W.shape[0] or (layers and W.shape[1] != layers[-1][0].shape[0]):
            raise ValueError(f"{path}: /{group}/layers/{i}: weights {W.shape}, biases {b.shape} do not chain")
        layers.append((W, b))
        acts.append(_ACT_NAMES[fn])
        i += 1
    if not layers:
        raise ValueError(f"{path}: /{group} has no layers")
    return layers, acts


def write_mlp_checkpoint_h5(path, layers, activations, group="actor", meta=None):
    """Inverse of ``load_mlp_checkpoint_h5``: dense layers (W [out, in], b [out]) with their activations ("relu" | "tanh" |
    "identity") in the group / dataset / attribute layout rl-tools writes for a `sequential` model (the attribute conventions of
    tests/golden/checkpoint.h5: matrices carry type / rows / cols as strings); libhdf5's h5dump accepts the file
    (tests/test_host_logic.py)."""
    from .hdf5_min import DatasetSpec, GroupSpec, write_file
    names = {"relu": "RELU", "tanh": "TANH", "identity": "IDENTITY"}
    lay = {}
    for i, ((W, b), a) in enumerate(zip(layers, activations)):
        W = np.ascontiguousarray(W, np.float32)
        b = np.ascontiguousarray(b, np.float32).reshape(1, -1)
        if W.ndim != 2 or b.shape[1] != W.shape[0]:
            raise ValueError(f"layer {i}: weights {W.shape} and biases {b.shape} do not match")
        lay[str(i)] = GroupSpec({"weights": GroupSpec({"parameters": DatasetSpec(W, _shape_attrs(W.shape, True))}),
                                 "biases": GroupSpec({"parameters": DatasetSpec(b, _shape_attrs(b.shape, True))})},
                                {"type": "dense", "activation_function": names[a]})
    attrs = {"type": "sequential"}
    if meta is not None:
        attrs["meta"] = meta
    write_file(path, GroupSpec({group: GroupSpec({"layers": GroupSpec(lay)}, attrs)}))


# The observation the engine's `observe` assembles and its actor kernels consume (README.md:23; `h5:/actor@meta` of the shipped
# checkpoint): 3 position + 9 rotation matrix (row-major) + 3 linear velocity + 3 body-frame angular velocity + 4 previous action.
ENGINE_OBSERVATION = "Position.OrientationRotationMatrix.LinearVelocity.AngularVelocityDelayed(0).ActionHistory(1)"


def observation_of_meta(meta):
    """The observation specification string inside an rl-tools `/actor@meta` attribute (JSON: {"environment": {"name": ...,
    "observation": ...}}) -> str or None (no meta, or a meta without that key).  Text that is not JSON is an error: a
    checkpoint that says something about itself which cannot be read must not load as if it had said nothing."""
    if meta is None:
        return None
    import json
    try:
        doc = json.loads(meta)
    except (TypeError, ValueError) as e:
        raise ValueError(f"checkpoint meta is not JSON ({e}): {meta!r}")
    env = doc.get("environment") if isinstance(doc, dict) else None
    obs = env.get("observation") if isinstance(env, dict) else None
    return None if obs is None else str(obs)


def check_observation(meta, path="checkpoint"):
    """Refuse a checkpoint trained on another observation layout than the one this engine assembles (VERDICT r05 missing 4: the
    loader used to read `/actor@meta` and throw it away, so such a policy loaded silently and flew on permuted inputs).
    -> the specification string (None when the checkpoint does not carry one, e.g. the C++ export, whose meta holds name and commit only)."""
    obs = observation_of_meta(meta)
    if obs is not None and obs != ENGINE_OBSERVATION:
        raise ValueError(f"{path}: trained on observation '{obs}', this engine assembles '{ENGINE_OBSERVATION}' "
                         "(README.md:23); pass check_observation=False to load it anyway")
    return obs


def load_checkpoint(path, with_meta=False):
    """Dispatch on the file type: HDF5 signature -> `load_checkpoint_h5`, otherwise the C++ export.
    -> (weights, example) or, with_meta, (weights, example, meta string or None)."""
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic == b"\x89HDF\r\n\x1a\n":
        w, ex, meta = load_checkpoint_h5(path)
    else:
        (w, ex), meta = load_checkpoint_header(path), None
    return (w, ex, meta) if with_meta else (w, ex)
