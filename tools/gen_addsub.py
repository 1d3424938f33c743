#!/usr/bin/env python3
"""Generates fawkes-crypto_amd/csrc/addsub_gfx950.inc: modular addition / subtraction of 8 x u32 limb field
elements as real carry chains.

hipcc does not emit carry chains for the C loop in field.hpp (a VALU instruction that reads a carry written by the
previous VALU instruction needs 2 wait states on gfx90a+/gfx950, so it falls back to 64-bit adds and moves: 91 VALU
instructions per addition, 84 per subtraction -- a quarter of a Montgomery product).  Here every operation is two
chains: the primary one (a + b, or a - b) and a secondary one running one step behind it on the primary's output
(the trial subtraction of p, or the corrective addition of p); the final borrow selects between them with
v_cndmask.  Carries live in SGPR pairs (VOP3 forms), the modulus limbs in VGPRs (gfx9 allows ONE scalar operand per
VALU instruction and the carry-in already is one).

HAZARD RULE (same as tools/gen_mont_mul.py): >= 2 instructions between the VALU write of an SGPR pair and a VALU read
of it.  The dual form (two independent operations = four chains, one limb of each per asm statement) satisfies it
with no padding: 4 instructions per limb pair.  The single form pads with one s_nop per limb.  `check()` replays the
generated statements back to back (assuming NO padding between statements) and verifies the rule.

Forms generated (members of Fp):  as1_A, as1_S (single add / sub);  as2_AA, as2_SS, as2_AS (r1 = a op1 b, r2 = c op2 d);
red2 (two conditional subtractions of p: the tail of the dual Montgomery product).
"""
import os
import re

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'fawkes-crypto_amd', 'csrc', 'addsub_gfx950.inc')


def prim(op, k, x, carry, a, b):
    m = {'A': ('v_add_co_u32', 'v_addc_co_u32'), 'S': ('v_sub_co_u32', 'v_subb_co_u32')}[op]
    if k == 0:
        return '%s %s, %s, %s, %s' % (m[0], x, carry, a, b)
    return '%s %s, %s, %s, %s, %s' % (m[1], x, carry, a, b, carry)


def sec(op, k, t, carry, x, p):
    # after an addition: trial subtraction of p; after a subtraction: corrective addition of p
    return prim('S' if op == 'A' else 'A', k, t, carry, x, p)


def select(op, r, x, t, mask):
    # A: secondary borrow set  <=> a + b < p  -> keep the primary sum;  S: primary borrow set <=> a < b -> take a - b + p
    if op == 'A':
        return 'v_cndmask_b32_e64 %s, %s, %s, %s' % (r, t, x, mask)
    return 'v_cndmask_b32_e64 %s, %s, %s, %s' % (r, x, t, mask)


def gen_single(o, op):
    o.append('    static __device__ __forceinline__ Fp as1_%s(const Fp &a, const Fp &b) {' % op)
    o.append('        uint32_t x[8], t[8]; uint64_t c, w; Fp r;')
    for k in range(8):
        cs = '"=&s"(c), "=&s"(w)' if k == 0 else '"+s"(c), "+s"(w)'
        body = '\\n\\t'.join([prim(op, k, '%0', '%2', '%4', '%5'), sec(op, k, '%1', '%3', '%0', '%6'), 's_nop 0'])
        o.append('        asm("%s" : "=&v"(x[%d]), "=&v"(t[%d]), %s : "v"(a.v[%d]), "v"(b.v[%d]), "v"(P::q(%d)));' % (body, k, k, cs, k, k, k))
    mask = 'w' if op == 'A' else 'c'
    for half in range(2):
        ks = range(4 * half, 4 * half + 4)
        lines = (['s_nop 0'] if half == 0 else []) + [select(op, '%%%d' % i, '%%%d' % (4 + 2 * i), '%%%d' % (5 + 2 * i), '%12') for i in range(4)]
        outs = ', '.join('"=&v"(r.v[%d])' % k for k in ks)
        ins = ', '.join('"v"(x[%d]), "v"(t[%d])' % (k, k) for k in ks)
        o.append('        asm("%s" : %s : %s, "s"(%s));' % ('\\n\\t'.join(lines), outs, ins, mask))
    o.append('        return r;')
    o.append('    }')


def gen_dual(o, op1, op2):
    o.append('    static __device__ __forceinline__ void as2_%s%s(const Fp &a, const Fp &b, const Fp &c, const Fp &d, Fp &r1, Fp &r2) {' % (op1, op2))
    o.append('        uint32_t x1[8], x2[8], t1[8], t2[8]; uint64_t c1, c2, w1, w2; Fp u, v;')
    for k in range(8):
        cs = ', '.join('"%s"(%s)' % ('=&s' if k == 0 else '+s', n) for n in ('c1', 'c2', 'w1', 'w2'))
        body = '\\n\\t'.join([prim(op1, k, '%0', '%4', '%8', '%9'), prim(op2, k, '%1', '%5', '%10', '%11'),
                              sec(op1, k, '%2', '%6', '%0', '%12'), sec(op2, k, '%3', '%7', '%1', '%12')])
        o.append('        asm("%s" : "=&v"(x1[%d]), "=&v"(x2[%d]), "=&v"(t1[%d]), "=&v"(t2[%d]), %s : "v"(a.v[%d]), "v"(b.v[%d]), "v"(c.v[%d]), "v"(d.v[%d]), "v"(P::q(%d)));'
                 % (body, k, k, k, k, cs, k, k, k, k, k))
    first = True
    for op, x, t, res, c, w in ((op1, 'x1', 't1', 'u', 'c1', 'w1'), (op2, 'x2', 't2', 'v', 'c2', 'w2')):
        mask = w if op == 'A' else c
        for half in range(2):
            ks = range(4 * half, 4 * half + 4)
            lines = (['s_nop 0'] if first else []) + [select(op, '%%%d' % i, '%%%d' % (4 + 2 * i), '%%%d' % (5 + 2 * i), '%12') for i in range(4)]
            first = False
            outs = ', '.join('"=&v"(%s.v[%d])' % (res, k) for k in ks)
            ins = ', '.join('"v"(%s[%d]), "v"(%s[%d])' % (x, k, t, k) for k in ks)
            o.append('        asm("%s" : %s : %s, "s"(%s));' % ('\\n\\t'.join(lines), outs, ins, mask))
    o.append('        r1 = u; r2 = v;')
    o.append('    }')


def gen_red2(o, name='red2', mod='p'):
    """two conditional subtractions of p at once (the tail of a dual Montgomery product: x, y < 2p); red2q: of q"""
    o.append('    static __device__ __forceinline__ void %s(const Fp &x, const Fp &y, Fp &r1, Fp &r2) {' % name)
    o.append('        uint32_t t1[8], t2[8]; uint64_t w1, w2; Fp u, v;')
    for k in range(8):
        cs = ', '.join('"%s"(%s)' % ('=&s' if k == 0 else '+s', n) for n in ('w1', 'w2'))
        body = '\\n\\t'.join([prim('S', k, '%0', '%2', '%4', '%6'), prim('S', k, '%1', '%3', '%5', '%6'), 's_nop 0'])
        o.append('        asm("%s" : "=&v"(t1[%d]), "=&v"(t2[%d]), %s : "v"(x.v[%d]), "v"(y.v[%d]), "v"(P::%s(%d)));' % (body, k, k, cs, k, k, mod, k))
    first = True
    for src, t, res, w in (('x', 't1', 'u', 'w1'), ('y', 't2', 'v', 'w2')):
        for half in range(2):
            ks = range(4 * half, 4 * half + 4)
            lines = (['s_nop 0'] if first else []) + ['v_cndmask_b32_e64 %%%d, %%%d, %%%d, %%12' % (i, 5 + 2 * i, 4 + 2 * i) for i in range(4)]
            first = False
            outs = ', '.join('"=&v"(%s.v[%d])' % (res, k) for k in ks)
            ins = ', '.join('"v"(%s.v[%d]), "v"(%s[%d])' % (src, k, t, k) for k in ks)
            o.append('        asm("%s" : %s : %s, "s"(%s));' % ('\\n\\t'.join(lines), outs, ins, w))
    o.append('        r1 = u; r2 = v;')
    o.append('    }')


def gen_red1q(o):
    """one conditional subtraction of q (the tail of a*b + c*d with one reduction)"""
    o.append('    static __device__ __forceinline__ Fp red1q(const Fp &x) {')
    o.append('        uint32_t t[8]; uint64_t w; Fp r;')
    for k in range(8):
        cs = '"%s"(w)' % ('=&s' if k == 0 else '+s')
        body = '\\n\\t'.join([prim('S', k, '%0', '%1', '%2', '%3'), 's_nop 0', 's_nop 0'])
        o.append('        asm("%s" : "=&v"(t[%d]), %s : "v"(x.v[%d]), "v"(P::q(%d)));' % (body, k, cs, k, k))
    for half in range(2):
        ks = range(4 * half, 4 * half + 4)
        lines = (['s_nop 0'] if half == 0 else []) + ['v_cndmask_b32_e64 %%%d, %%%d, %%%d, %%12' % (i, 5 + 2 * i, 4 + 2 * i) for i in range(4)]
        outs = ', '.join('"=&v"(r.v[%d])' % k for k in ks)
        ins = ', '.join('"v"(x.v[%d]), "v"(t[%d])' % (k, k) for k in ks)
        o.append('        asm("%s" : %s : %s, "s"(w));' % ('\\n\\t'.join(lines), outs, ins))
    o.append('        return r;')
    o.append('    }')


def check(text):
    """Replay each function's asm statements back to back; operands are resolved to their C names so that carries are
    tracked across statements.  Every read of an SGPR carry / mask must be >= 2 instructions after its last write."""
    bad = n = 0
    for fn in re.finditer(r'static __device__ __forceinline__ [^\n]*\{\n(.*?)\n    \}', text, re.S):
        last, pos = {}, 0
        for st in re.finditer(r'asm\("(.*?)" : (.*?) : (.*?)\);', fn.group(1)):
            names = [m.group(1) for m in re.finditer(r'"[^"]*"\(([^()]*(?:\([^()]*\))?[^()]*)\)', st.group(2) + ', ' + st.group(3))]
            for ins in st.group(1).split('\\n\\t'):
                tk = ins.replace(',', ' ').split()
                ops = [names[int(x[1:])] if x.startswith('%') else x for x in tk[1:]]
                if tk[0].startswith(('v_add', 'v_sub')):
                    if len(ops) == 5:                      # carry-in form: reads ops[4]
                        n += 1
                        if ops[4] in last and pos - last[ops[4]] - 1 < 2:
                            bad += 1
                    last[ops[1]] = pos
                elif tk[0].startswith('v_cndmask'):
                    n += 1
                    if ops[3] in last and pos - last[ops[3]] - 1 < 2:
                        bad += 1
                pos += 1
    return n, bad


def main():
    o = ['// GENERATED by tools/gen_addsub.py -- do not edit.',
         '// Modular addition / subtraction as carry chains (primary chain + lagging correction chain + select), gfx950',
         '// inline asm; see the generator for the schedule and the SGPR-carry hazard rule it obeys.']
    gen_single(o, 'A')
    gen_single(o, 'S')
    gen_dual(o, 'A', 'A')
    gen_dual(o, 'S', 'S')
    gen_dual(o, 'A', 'S')
    gen_red2(o)
    gen_red2(o, 'red2q', 'q')
    gen_red1q(o)
    text = '\n'.join(o) + '\n'
    n, bad = check(text)
    if bad or n == 0:
        raise SystemExit('hazard rule violated in %d of %d carry / mask reads' % (bad, n))
    open(OUT, 'w').write(text)
    print('wrote %s (%d carry / mask reads checked against the SGPR hazard rule, 0 violations)' % (os.path.normpath(OUT), n))


if __name__ == '__main__':
    main()
