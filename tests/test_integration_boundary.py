"""The drop-in boundary, machine-checked.

(a) Every `extern "C"` prototype in INTEGRATION.md's Rust blocks is compared with the C declaration of the same name in
    include/garage_ec.h / include/garage_block.h: arity, and per argument (and for the return value) the pointer depth and
    the integer width / signedness.  A prototype that drifts from the header turns this test red
    (test_a_broken_prototype_is_caught proves the checker can see it).
(b) Every `src/...rs:LINE[-LINE]` anchor cited in INTEGRATION.md and the two headers must exist in /root/reference with
    that many lines, and the anchors the patch depends on must still hold the identifier they are cited for.  Skipped
    cleanly where /root/reference is absent (the GPU box)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HEADERS = [os.path.join(ROOT, "include", "garage_ec.h"), os.path.join(ROOT, "include", "garage_block.h")]

# ---------------------------------------------------------------- C side
C_INT = {
    "int": ("i", 32), "int32_t": ("i", 32), "uint32_t": ("u", 32), "uint64_t": ("u", 64), "int64_t": ("i", 64),
    "uint8_t": ("u", 8), "uint16_t": ("u", 16), "size_t": ("u", "size"), "char": ("i", 8), "double": ("f", 64),
}


def _strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def c_prototypes():
    """name -> (ret_shape, [arg_shapes]); shape = ("ptr", depth, base) | ("int", sign, width) | ("void",)"""
    protos = {}
    for h in HEADERS:
        src = _strip_comments(open(h).read())
        src = re.sub(r"#[^\n]*", " ", src)
        # function-pointer typedefs count as opaque pointer types
        fn_typedefs = set(re.findall(r"typedef\s+[^;]*?\(\s*\*\s*(\w+)\s*\)\s*\([^;]*?\)\s*;", src, flags=re.S))
        src = re.sub(r"typedef\s+[^;]*?\(\s*\*\s*\w+\s*\)\s*\([^;]*?\)\s*;", " ", src, flags=re.S)
        struct_names = set(re.findall(r"typedef\s+struct\s+\w*\s*(?:\{[^}]*\})?\s*(\w+)\s*;", src, flags=re.S))
        struct_names |= set(re.findall(r"typedef\s+struct\s+(\w+)\s+\w+\s*;", src))
        for m in re.finditer(r"([\w\s\*]+?)\b(g(?:ec|bm)_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
            ret, name, args = m.group(1), m.group(2), m.group(3)
            if "typedef" in ret:
                continue
            arg_list = [] if args.strip() in ("", "void") else _split_args(args)
            protos[name] = (_c_shape(ret, fn_typedefs, struct_names), [_c_shape(_drop_name(a, fn_typedefs | struct_names), fn_typedefs, struct_names) for a in arg_list])
    return protos


def _split_args(args):
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return [a.strip() for a in out]


TYPE_WORDS = set(C_INT) | {"void", "unsigned", "signed", "long", "short", "float"}


def _drop_name(arg, known_types=()):
    """'const uint8_t *const *blocks' -> 'const uint8_t *const *'; 'uint8_t hash[32]' -> 'uint8_t *'; 'unsigned n' -> 'unsigned'"""
    arg = re.sub(r"\[[^\]]*\]", " * ", arg)            # an array parameter decays to a pointer
    toks = re.findall(r"\w+|\*", arg)
    idents = [i for i, t in enumerate(toks) if t not in ("*", "const", "struct")]
    if len(idents) >= 2 and toks[idents[-1]] not in TYPE_WORDS and toks[idents[-1]] not in known_types:
        name_at = idents[-1]
        toks = toks[:name_at] + toks[name_at + 1:]
    return " ".join(toks)


def _c_shape(decl, fn_typedefs, struct_names):
    depth = decl.count("*")
    words = [w for w in re.findall(r"\w+", decl) if w not in ("const", "struct", "extern")]
    base = words[-1] if words else "void"
    if base in fn_typedefs:
        return ("ptr", depth + 1, "fn")
    if "unsigned" in words and base in ("unsigned", "int"):
        kind = ("u", 32)
    elif base in C_INT:
        kind = C_INT[base]
    else:
        kind = None
    if depth:
        return ("ptr", depth, "int" if kind else ("void" if base == "void" else "opaque"))
    if base == "void":
        return ("void",)
    if kind:
        return ("int",) + kind
    raise AssertionError(f"unknown C type in {decl!r}")


# ---------------------------------------------------------------- Rust side
RUST_INT = {"c_int": ("i", 32), "i32": ("i", 32), "u32": ("u", 32), "u64": ("u", 64), "i64": ("i", 64), "u8": ("u", 8),
            "u16": ("u", 16), "usize": ("u", "size"), "c_char": ("i", 8), "f64": ("f", 64)}


def rust_prototypes(text):
    protos = {}
    for block in re.findall(r"```rust(.*?)```", text, flags=re.S):
        for ext in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', block, flags=re.S):
            body = re.sub(r"//[^\n]*", " ", ext.group(1))
            for m in re.finditer(r"\bfn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", body, flags=re.S):
                name, args, ret = m.group(1), m.group(2), m.group(3)
                arg_list = [a.strip() for a in _split_args(args) if a.strip()]
                shapes = [_rust_shape(a.split(":", 1)[1]) for a in arg_list]
                protos[name] = (_rust_shape(ret) if ret else ("void",), shapes)
    return protos


def _rust_shape(t):
    t = t.strip()
    depth = t.count("*")
    if "extern" in t or t.startswith("Option<"):   # Option<unsafe extern "C" fn(..)>: a nullable function pointer
        return ("ptr", 1, "fn")
    base = re.findall(r"\w+", t)[-1]
    if depth:
        return ("ptr", depth, "int" if base in RUST_INT else ("void" if base == "c_void" else "opaque"))
    if base in RUST_INT:
        return ("int",) + RUST_INT[base]
    raise AssertionError(f"unknown Rust type {t!r}")


def _compatible(c, r):
    if c[0] != r[0]:
        return False
    if c[0] == "ptr":
        # depth must match; a pointee that is an integer on one side must be one on the other (void* <-> *mut c_void,
        # opaque struct <-> opaque struct)
        return c[1] == r[1] and c[2] == r[2]
    return c == r


def mismatches(rust, c):
    bad = []
    for name, (rret, rargs) in sorted(rust.items()):
        if name not in c:
            bad.append(f"{name}: not declared in include/*.h")
            continue
        cret, cargs = c[name]
        if len(cargs) != len(rargs):
            bad.append(f"{name}: {len(rargs)} arguments in INTEGRATION.md, {len(cargs)} in the header")
            continue
        if not _compatible(cret, rret):
            bad.append(f"{name}: return {rret} vs header {cret}")
        for i, (ca, ra) in enumerate(zip(cargs, rargs)):
            if not _compatible(ca, ra):
                bad.append(f"{name}: argument {i} is {ra} in INTEGRATION.md, {ca} in the header")
    return bad


def test_rust_prototypes_match_the_headers():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rust = rust_prototypes(text)
    c = c_prototypes()
    assert len(rust) >= 24, sorted(rust)
    assert "gec_encode_batch" in rust and "gec_codec_create" in rust
    # the parser really sees the header: every symbol the ctypes layer lists has a parsed prototype
    from garage_amd import _lib

    assert not [s for s in _lib.SYMBOLS if s not in c], [s for s in _lib.SYMBOLS if s not in c]
    bad = mismatches(rust, c)
    assert not bad, "\n".join(bad)


def test_a_broken_prototype_is_caught():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    c = c_prototypes()
    assert not mismatches(rust_prototypes(text), c)
    # one argument fewer
    broken = text.replace("fn gec_shard_len(k: c_int, block_len: usize) -> usize;", "fn gec_shard_len(k: c_int) -> usize;")
    assert broken != text and any("gec_shard_len" in b for b in mismatches(rust_prototypes(broken), c))
    # an integer where the header has a pointer
    broken = text.replace("fn gec_codec_destroy(c: *mut gec_codec);", "fn gec_codec_destroy(c: usize);")
    assert broken != text and any("gec_codec_destroy" in b for b in mismatches(rust_prototypes(broken), c))
    # the wrong integer width
    broken = text.replace("fn gec_host_alloc(bytes: usize) -> *mut c_void;", "fn gec_host_alloc(bytes: u32) -> *mut c_void;")
    assert broken != text and any("gec_host_alloc" in b for b in mismatches(rust_prototypes(broken), c))
    # a pointer level lost
    broken = text.replace("blocks: *const *const u8,\n                        block_len", "blocks: *const u8,\n                        block_len")
    assert broken != text and any("gec_encode_batch" in b for b in mismatches(rust_prototypes(broken), c))


# ---------------------------------------------------------------- anchors into /root/reference
# (path, first line, last line, identifier that must appear in [first-2, last+2])
ANCHORS = [
    ("src/block/manager.rs", 366, 408, "rpc_put_block"),
    ("src/block/manager.rs", 375, 378, "from_buffer"),
    ("src/block/manager.rs", 276, 339, "rpc_get_raw_block_internal"),
    ("src/block/manager.rs", 344, 363, "rpc_get_block_streaming"),
    ("src/block/manager.rs", 54, 73, "BlockRpc"),
    ("src/block/manager.rs", 577, 609, "read_block_from"),
    ("src/block/manager.rs", 627, 662, "find_block"),
    ("src/block/manager.rs", 720, 805, "write_block_inner"),
    ("src/block/manager.rs", 122, 192, "fn new"),
    ("src/block/block.rs", 85, 96, "from_buffer"),
    ("src/block/block.rs", 69, 83, "verify"),
    ("src/block/block.rs", 99, 106, "zstd_encode"),
    ("src/rpc/rpc_helper.rs", 432, 538, "try_write_many_sets"),
    ("src/rpc/rpc_helper.rs", 323, 411, "try_call_many_inner"),
    ("src/rpc/rpc_helper.rs", 570, 660, "block_read_nodes_of"),
    ("src/block/resync.rs", 354, 503, "resync_block"),
    ("src/block/resync.rs", 485, 499, "rpc_get_raw_block"),
    ("src/block/repair.rs", 450, 458, "read_block"),
    ("src/util/data.rs", 130, 138, "blake2sum"),
    ("src/rpc/layout/version.rs", 101, 104, "partition_of"),
    ("src/api/s3/put.rs", 42, 42, "PUT_BLOCKS_MAX_PARALLEL"),
    ("src/api/s3/get.rs", 429, 429, "mpsc::channel::<ByteStream>(2)"),
    ("src/net/message.rs", 229, 245, "clone"),
    ("src/util/error.rs", 70, 77, "CorruptData"),
    ("src/block/lib.rs", 13, 13, "zstd_encode"),
]

needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference is not on this box")


@needs_reference
@pytest.mark.parametrize("path,lo,hi,ident", ANCHORS, ids=[f"{a[0]}:{a[1]}:{a[3]}" for a in ANCHORS])
def test_anchor_still_says_what_the_patch_assumes(path, lo, hi, ident):
    lines = open(os.path.join(REF, path), errors="replace").read().splitlines()
    assert hi <= len(lines), f"{path} has {len(lines)} lines, cited {lo}-{hi}"
    window = "\n".join(lines[max(0, lo - 3):hi + 2])
    assert ident in window, f"{path}:{lo}-{hi} no longer mentions {ident!r}"


@needs_reference
def test_every_cited_line_exists():
    cited = set()
    for f in HEADERS + [os.path.join(ROOT, "INTEGRATION.md")]:
        cited |= set(re.findall(r"(src/[a-z_0-9/]+\.rs):(\d+)(?:-(\d+))?", open(f).read()))
    assert len(cited) > 50
    bad = []
    nlines = {}
    for path, lo, hi in sorted(cited):
        full = os.path.join(REF, path)
        if not os.path.exists(full):
            bad.append(f"{path}: no such file")
            continue
        if path not in nlines:
            nlines[path] = len(open(full, errors="replace").read().splitlines())
        last = int(hi) if hi else int(lo)
        if last > nlines[path] or (hi and int(hi) < int(lo)):
            bad.append(f"{path}:{lo}-{hi}: file has {nlines[path]} lines")
    assert not bad, "\n".join(bad)


def test_anchor_table_covers_the_patch():
    """every anchor of the table is one INTEGRATION.md or a header actually cites"""
    text = "".join(open(f).read() for f in HEADERS + [os.path.join(ROOT, "INTEGRATION.md"), os.path.join(ROOT, "SURVEY.md")] if os.path.exists(f))
    for path, lo, hi, _ in ANCHORS:
        short = path.split("/")[-1]
        assert re.search(re.escape(short) + r":" + str(lo) + r"\b", text) or re.search(r":" + str(lo) + r"-" + str(hi) + r"\b", text), (path, lo, hi)
