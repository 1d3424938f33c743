"""The Rust binding is checkable without Rust: `shim/src/ffi.rs` and the `extern "C"` block of INTEGRATION.md section 2 are
hand-written mirrors of include/fawkes_hip.h.  This test parses all three and fails on any difference in a function's name, arity,
pointer / integer kind, pointee type or constness, and in the field order / widths of `fk_key_desc`; it compiles a C99 stub that prints
sizeof / offsetof of the header's structs and compares them with the `#[repr(C)]` layout computed here from the Rust declaration and with
the ctypes structures of fawkes-crypto_amd/api.py.  Reference interface: prover.rs:63-68 (what the shim's `prove` wraps), SURVEY 8(b)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'fawkes_hip.h')

OPAQUE = {'fk_ctx', 'fk_key', 'fk_r1cs_dev', 'fk_gates', 'fk_multi', 'fk_multi_key', 'fk_multi_r1cs', 'fk_blob'}
STRUCTS = {'fk_key_desc', 'fk_timings', 'fk_r1cs'}
C_SCALARS = {'int': 'i32', 'int32_t': 'i32', 'uint32_t': 'u32', 'uint64_t': 'u64', 'size_t': 'usize', 'double': 'f64', 'uint8_t': 'u8', 'char': 'char', 'void': 'void', 'unsigned': 'u32'}
RS_SCALARS = {'c_int': 'i32', 'i32': 'i32', 'u32': 'u32', 'u64': 'u64', 'usize': 'usize', 'f64': 'f64', 'u8': 'u8', 'c_char': 'char', 'c_void': 'void',
              'std::os::raw::c_char': 'char', 'std::ffi::c_void': 'void', 'std::os::raw::c_void': 'void'}
# INTEGRATION.md spells the opaque types in CamelCase
CAMEL = {'FkCtx': 'fk_ctx', 'FkKey': 'fk_key', 'FkKeyDesc': 'fk_key_desc', 'FkR1cs': 'fk_r1cs', 'FkR1csDev': 'fk_r1cs_dev', 'FkGates': 'fk_gates', 'FkMulti': 'fk_multi',
         'FkMultiKey': 'fk_multi_key', 'FkMultiR1cs': 'fk_multi_r1cs'}


def _strip_c(text):
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    text = re.sub(r'//[^\n]*', ' ', text)
    return '\n'.join(l for l in text.splitlines() if not l.strip().startswith('#'))


def _c_type(t):
    """'const uint64_t *' -> ('ptr', const?, pointee) | ('val', scalar)"""
    t = t.strip()
    depth = t.count('*')
    base = t.replace('*', ' ')
    const = bool(re.search(r'\bconst\b', base))
    base = re.sub(r'\b(const|struct)\b', ' ', base).split()
    base = ' '.join(base)
    name = C_SCALARS.get(base, base)
    if depth == 0:
        return ('val', name)
    if depth == 1:
        return ('ptr', const, name)
    return ('ptr', False, ('ptr', const, name))


def parse_header():
    text = _strip_c(open(HEADER).read())
    structs = {}
    for m in re.finditer(r'typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;', text, flags=re.S):
        fields = []
        for decl in m.group(1).split(';'):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r'(.*?)([\w\s,\*\[\]]+)$', decl, flags=re.S)
            # "const uint8_t *alpha_g1, *beta_g1" / "uint64_t m" / "double ms[4]"
            first = re.match(r'((?:const\s+)?\w+)\s*(.*)$', decl, flags=re.S)
            base, rest = first.group(1), first.group(2)
            for item in rest.split(','):
                item = item.strip()
                arr = re.search(r'\[(\w+)\]', item)
                nm = re.sub(r'[\*\[\]\w]*\[.*', '', item) if arr else item
                nm = re.sub(r'\[.*', '', item).replace('*', '').strip()
                ty = _c_type(base + ' ' + '*' * item.count('*'))
                fields.append((nm, ty, int(arr.group(1)) if arr and arr.group(1).isdigit() else (arr.group(1) if arr else None)))
        structs[m.group(2)] = fields
    text_nostruct = re.sub(r'typedef\s+struct\s*\{.*?\}\s*\w+\s*;', ' ', text, flags=re.S)
    funcs = {}
    for m in re.finditer(r'([\w\s\*]+?)\b(fk_\w+)\s*\(([^;{}]*?)\)\s*;', text_nostruct, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith('typedef') or not ret:
            continue
        params = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                arr = '[' in a
                a = re.sub(r'\[.*?\]', '', a)
                mm = re.match(r'(.*?)(\w+)$', a.strip(), flags=re.S)
                ty, nm = mm.group(1), mm.group(2)
                params.append((nm, _c_type(ty + ('*' if arr else ''))))
        funcs[name] = (_c_type(ret), params)
    return funcs, structs


def _rs_type(t):
    t = t.strip()
    m = re.match(r'\*(const|mut)\s+(.*)$', t)
    if m:
        inner = _rs_type(m.group(2))
        const = m.group(1) == 'const'
        if inner[0] == 'ptr':
            return ('ptr', False, inner) if not const else ('ptr', True, inner)
        return ('ptr', const, inner[1])
    name = RS_SCALARS.get(t, CAMEL.get(t, t))
    return ('val', name)


def parse_rust(text):
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    text = re.sub(r'//[^\n]*', ' ', text)
    funcs = {}
    for m in re.finditer(r'pub\s+fn\s+(fk_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;', text, flags=re.S):
        params = []
        for a in m.group(2).split(','):
            a = a.strip()
            if not a:
                continue
            nm, ty = a.split(':', 1)
            params.append((nm.strip(), _rs_type(ty)))
        funcs[m.group(1)] = (_rs_type(m.group(3)) if m.group(3) else ('val', 'void'), params)
    structs = {}
    for m in re.finditer(r'#\[repr\(C\)\]\s*pub\s+struct\s+(\w+)\s*\{(.*?)\}', text, flags=re.S):
        body = m.group(2)
        if '_p:' in body:
            continue
        fields = []
        for f in body.split(','):
            f = f.strip()
            if not f:
                continue
            nm, ty = f.replace('pub ', '').split(':', 1)
            fields.append((nm.strip(), _rs_type(ty)))
        structs[CAMEL.get(m.group(1), m.group(1))] = fields
    return funcs, structs


def _same_kind(c, r):
    """C type vs Rust type; `c_void` on the Rust side stands for any pointee (opaque `timings`)"""
    if c[0] != r[0]:
        return False
    if c[0] == 'val':
        return c[1] == r[1]
    if c[1] != r[1]:                      # constness
        return False
    if isinstance(c[2], tuple) or isinstance(r[2], tuple):
        return isinstance(c[2], tuple) and isinstance(r[2], tuple) and _same_kind(c[2], r[2])
    return r[2] == 'void' or c[2] == r[2]


def _check_mirror(rs_funcs, rs_structs, where):
    c_funcs, c_structs = parse_header()
    assert len(c_funcs) > 100 and 'fk_prove_r1cs' in c_funcs and 'fk_key_desc' in c_structs
    assert rs_funcs, 'no extern "C" declarations found in ' + where
    problems = []
    for name, (ret, params) in rs_funcs.items():
        if name not in c_funcs:
            problems.append('%s: %s is not declared in include/fawkes_hip.h' % (where, name))
            continue
        c_ret, c_params = c_funcs[name]
        if not _same_kind(c_ret, ret):
            problems.append('%s: %s returns %s, the header says %s' % (where, name, ret, c_ret))
        if len(params) != len(c_params):
            problems.append('%s: %s takes %d arguments, the header %d' % (where, name, len(params), len(c_params)))
            continue
        for (rn, rt), (cn, ct) in zip(params, c_params):
            if not _same_kind(ct, rt):
                problems.append('%s: %s argument %s is %s, the header has %s %s' % (where, name, rn, rt, cn, ct))
    for sname, fields in rs_structs.items():
        if sname not in c_structs:
            problems.append('%s: struct %s is not in the header' % (where, sname))
            continue
        cf = c_structs[sname]
        if [f[0] for f in cf] != [f[0] for f in fields]:
            problems.append('%s: struct %s fields %s, the header has %s' % (where, sname, [f[0] for f in fields], [f[0] for f in cf]))
            continue
        for (cn, ct, carr), (rn, rt) in zip(cf, fields):
            if carr is not None or not _same_kind(ct, rt):
                problems.append('%s: struct %s field %s is %s, the header has %s' % (where, sname, rn, rt, ct))
    assert not problems, '\n'.join(problems)
    return c_funcs, c_structs


def test_shim_ffi_rs_mirrors_the_header():
    rs_funcs, rs_structs = parse_rust(open(os.path.join(ROOT, 'shim', 'src', 'ffi.rs')).read())
    assert 'fk_key_desc' in rs_structs and len(rs_funcs) >= 25
    _check_mirror(rs_funcs, rs_structs, 'shim/src/ffi.rs')
    # every entry point shim/src/lib.rs calls is declared in ffi.rs
    used = set(re.findall(r'\bffi::(fk_\w+)\s*\(', open(os.path.join(ROOT, 'shim', 'src', 'lib.rs')).read()))
    assert used and used <= set(rs_funcs), sorted(used - set(rs_funcs))


def test_integration_md_extern_block_mirrors_the_header():
    md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = md[md.index('## 2. `extern "C"` block'):md.index('## 3. ')]
    code = '\n'.join(re.findall(r'```rust(.*?)```', sec, flags=re.S))
    rs_funcs, rs_structs = parse_rust(code)
    assert 'fk_key_desc' in rs_structs and len(rs_funcs) >= 15
    _check_mirror(rs_funcs, rs_structs, 'INTEGRATION.md section 2')


def test_the_parser_notices_a_wrong_mirror():
    """negative control: a swapped argument, a dropped const, a narrower integer and a reordered struct field are all reported"""
    good = open(os.path.join(ROOT, 'shim', 'src', 'ffi.rs')).read()
    for old, new in (('num_gates: u32, num_input: u32', 'num_gates: u64, num_input: u32'),
                     ('pub fn fk_prove_r1cs(ctx: *mut fk_ctx, key: *const fk_key', 'pub fn fk_prove_r1cs(ctx: *mut fk_ctx, key: *mut fk_key'),
                     ('pub fn fk_free(ctx: *mut fk_ctx);', 'pub fn fk_free(ctx: *mut fk_ctx, flags: u32);'),
                     ('pub h: *const u8, pub n_h: u64,', 'pub n_h: u64, pub h: *const u8,'),
                     ('pub fn fk_gates_free(gates: *mut fk_gates);', 'pub fn fk_gates_release(gates: *mut fk_gates);')):
        assert old in good, old
        f, s = parse_rust(good.replace(old, new))
        with pytest.raises(AssertionError):
            _check_mirror(f, s, 'mutated ffi.rs')


def _repr_c_layout(fields):
    """offsets / size / alignment of a #[repr(C)] struct on x86-64 (= the C ABI's rule)"""
    width = {'u8': 1, 'i32': 4, 'u32': 4, 'u64': 8, 'usize': 8, 'f64': 8}
    off, offs, align = 0, {}, 1
    for name, ty in fields:
        w = 8 if ty[0] == 'ptr' else width[ty[1]]
        off = (off + w - 1) // w * w
        offs[name] = off
        off += w
        align = max(align, w)
    return offs, (off + align - 1) // align * align


def test_struct_layouts_c_vs_rust_vs_ctypes():
    _, c_structs = parse_header()
    names = {s: [f[0] for f in c_structs[s]] for s in STRUCTS}
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "fawkes_hip.h"', 'int main(void) {']
    for s in sorted(STRUCTS):
        prog.append('  printf("%s size %%zu\\n", sizeof(%s));' % (s, s))
        for f in names[s]:
            prog.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (s, f, s, f))
    prog += ['  return 0;', '}']
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, 'layout.c'), os.path.join(td, 'layout')
        open(src, 'w').write('\n'.join(prog))
        subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), src, '-o', exe])
        out = subprocess.check_output([exe], text=True)
    c_layout = {}
    for line in out.splitlines():
        s, f, v = line.split()
        c_layout.setdefault(s, {})[f] = int(v)
    # Rust: shim/src/ffi.rs and INTEGRATION.md declare fk_key_desc
    for path, text in (('shim/src/ffi.rs', open(os.path.join(ROOT, 'shim', 'src', 'ffi.rs')).read()), ('INTEGRATION.md', open(os.path.join(ROOT, 'INTEGRATION.md')).read())):
        _, rs_structs = parse_rust(text)
        offs, size = _repr_c_layout(rs_structs['fk_key_desc'])
        want = dict(c_layout['fk_key_desc'])
        assert size == want.pop('size') and offs == want, (path, offs, c_layout['fk_key_desc'])
    # ctypes: what every GPU test goes through
    sys.path.insert(0, ROOT)
    from fawkes_crypto_amd import api
    for sname, cls in (('fk_key_desc', api.KeyDesc), ('fk_timings', api.Timings), ('fk_r1cs', api.R1csStruct)):
        want = dict(c_layout[sname])
        assert C.sizeof(cls) == want.pop('size'), sname
        got = {f[0]: getattr(cls, f[0]).offset for f in cls._fields_}
        assert got == want, (sname, got, want)
