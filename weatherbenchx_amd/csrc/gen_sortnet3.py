#!/usr/bin/env python3
"""Generates wbx_sortnet3_gen.hpp: sorting networks of compare-exchanges AND 3-sorters for the exact-M ensemble kernels.

On gfx950 `v_min3_f32 / v_med3_f32 / v_max3_f32` cost what `v_min_f32 / v_max_f32` do (tools/ubench/valu_ops.hip), so a
3-sorter is 3 instructions where the three compare-exchanges it replaces are 6.  Batcher's network for M = 51 is 415
compare-exchanges = 830 instructions, two thirds of the ensemble kernel's VALU time.  This generator
builds a merge sort instead:

  * leaves: cost-optimal networks of compare-exchanges (2 instructions) and 3-sorters (3) for n <= 7, found by exhaustive
    search over the reachable sets of 0-1 vectors (the sequences below; re-verified here over all 2^n inputs);
  * merges: Batcher's odd-even merge for any two lengths on wire LISTS (the output is a wire order, not a position:
    registers are renamed for free), pruned of compare-exchanges that never swap, then a peephole pass that fuses two
    compare-exchanges meeting on three wires into one 3-sorter wherever the merge still sorts -- every candidate is checked
    against ALL valid inputs of the merge by the 0-1 principle: two sorted 0-1 lists have (a+1)(b+1) forms, simulated
    together as bit vectors (min = AND, max = OR, median = majority);
  * the split of every size is chosen by dynamic programming over the instruction counts.

Correctness is compositional (every leaf exhaustively, every merge over all its valid 0-1 inputs) and the emitted network is
finally run on random vectors.  Usage: python gen_sortnet3.py > wbx_sortnet3_gen.hpp
"""
import functools
import itertools
import random
import sys

SIZES = [4, 8, 16, 32, 50, 51, 64]

# (op, wires...): ('ce', lo, hi) or ('s3', lo, mid, hi); found by the exhaustive search described above
LEAVES = {
    1: [],
    2: [('ce', 0, 1)],
    3: [('s3', 0, 1, 2)],
    4: [('ce', 0, 1), ('s3', 0, 2, 3), ('s3', 1, 2, 3)],
    5: [('s3', 0, 1, 2), ('ce', 1, 3), ('s3', 0, 1, 4), ('s3', 2, 3, 4)],
    6: [('ce', 0, 1), ('s3', 2, 3, 4), ('ce', 0, 3), ('s3', 1, 4, 5), ('s3', 0, 1, 2), ('s3', 2, 3, 4)],
    7: [('s3', 0, 1, 2), ('s3', 3, 4, 5), ('s3', 1, 4, 6), ('s3', 0, 1, 3), ('s3', 2, 5, 6), ('s3', 2, 3, 4)],
}


def cost(net):
  return sum(2 if op[0] == 'ce' else 3 for op in net)


def run(net, v):
  v = list(v)
  for op in net:
    if op[0] == 'ce':
      _, a, b = op
      if v[a] > v[b]:
        v[a], v[b] = v[b], v[a]
    else:
      _, a, b, c = op
      v[a], v[b], v[c] = sorted((v[a], v[b], v[c]))
  return v


for _n, _net in LEAVES.items():
  for _bits in itertools.product((0, 1), repeat=_n):
    _out = run(_net, _bits)
    assert _out == sorted(_out), (_n, _bits)


def oem(A, B):
  """Batcher's odd-even merge of the sorted wire lists A and B -> (compare-exchanges, output wire order) (Knuth 5.3.4)."""
  if not A:
    return [], list(B)
  if not B:
    return [], list(A)
  if len(A) == 1 and len(B) == 1:
    return [('ce', A[0], B[0])], [A[0], B[0]]
  ne, E = oem(A[0::2], B[0::2])
  no, O = oem(A[1::2], B[1::2])
  net, out, i = ne + no, [E[0]], 0
  while i < len(O) and i + 1 < len(E):
    net.append(('ce', O[i], E[i + 1]))
    out += [O[i], E[i + 1]]
    i += 1
  return net, out + O[i:] + E[i + 1:]


class MergeCheck:
  """All valid inputs of a merge of a + b sorted wires at once: bit t of wire w = value of w in the t-th 0-1 input."""

  def __init__(self, a, b):
    self.n = a + b
    forms = [(za, zb) for za in range(a + 1) for zb in range(b + 1)]
    self.full = (1 << len(forms)) - 1
    self.wires = []
    for w in range(a + b):
      bits = 0
      for t, (za, zb) in enumerate(forms):
        one = (w >= za) if w < a else (w - a >= zb)
        bits |= int(one) << t
      self.wires.append(bits)

  def simulate(self, net):
    """-> (final wire values, per-op 'did something' flags)."""
    v = list(self.wires)
    active = []
    for op in net:
      if op[0] == 'ce':
        _, a, b = op
        lo, hi = v[a] & v[b], v[a] | v[b]
        active.append(lo != v[a])
        v[a], v[b] = lo, hi
      else:
        _, a, b, c = op
        lo, hi = v[a] & v[b] & v[c], v[a] | v[b] | v[c]
        md = (v[a] & v[b]) | (v[a] & v[c]) | (v[b] & v[c])
        active.append((lo, md, hi) != (v[a], v[b], v[c]))
        v[a], v[b], v[c] = lo, md, hi
    return v, active

  def sorts(self, net, out):
    v, _ = self.simulate(net)
    # ascending along `out` for every input: a 1 is never followed by a 0
    return all(v[out[k]] & ~v[out[k + 1]] & self.full == 0 for k in range(len(out) - 1))

  def prune(self, net):
    _, active = self.simulate(net)
    return [op for op, act in zip(net, active) if act]


def peephole(chk, net, out):
  """Fuses pairs of compare-exchanges that meet on three wires into one 3-sorter while the merge still sorts."""
  improved = True
  while improved:
    improved = False
    for p, op in enumerate(net):
      if op[0] != 'ce':
        continue
      x, y = op[1], op[2]
      nxt = {}
      for q in range(p + 1, len(net)):
        for w in net[q][1:]:
          if w in (x, y) and w not in nxt:
            nxt[w] = q
        if len(nxt) == 2:
          break
      for w, other in ((x, y), (y, x)):
        q = nxt.get(w)
        if q is None or net[q][0] != 'ce' or nxt.get(other, len(net)) <= q:
          continue
        z = [u for u in net[q][1:] if u != w]
        if len(z) != 1 or z[0] in (x, y):
          continue
        for roles in itertools.permutations((x, y, z[0])):
          cand = net[:p] + net[p + 1:q] + [('s3',) + roles] + net[q + 1:]
          if chk.sorts(cand, out):
            net = chk.prune(cand)
            improved = True
            break
        if improved:
          break
      if improved:
        break
  return net


@functools.lru_cache(None)
def merge_net(a, b):
  """Optimised merge of sorted wires [0, a) and [a, a + b) -> (ops, output wire order)."""
  chk = MergeCheck(a, b)
  net, out = oem(list(range(a)), list(range(a, a + b)))
  assert chk.sorts(net, out), (a, b)
  net = peephole(chk, chk.prune(net), out)
  assert chk.sorts(net, out), (a, b)
  return net, out


@functools.lru_cache(None)
def plan(n):
  """-> (instructions, split or None): cheapest merge sort of n wires."""
  if n in LEAVES:  # (the searched leaves are optimal: no split beats them)
    return cost(LEAVES[n]), None
  best = None
  for k in range(1, n // 2 + 1):
    c = plan(k)[0] + plan(n - k)[0] + cost(merge_net(k, n - k)[0])
    if best is None or c < best[0]:
      best = (c, k)
  return best


def build(wires):
  """-> (ops on the given wires, sorted wire order)."""
  n = len(wires)
  split = plan(n)[1]
  if split is None:
    return [(op[0],) + tuple(wires[w] for w in op[1:]) for op in LEAVES[n]], list(wires)
  na, oa = build(wires[:split])
  nb, ob = build(wires[split:])
  m, out = merge_net(split, n - split)
  src = oa + ob  # merge wire t is the t-th element of (sorted A, sorted B)
  return na + nb + [(op[0],) + tuple(src[w] for w in op[1:]) for op in m], [src[w] for w in out]


def verify(n, net, out):
  rng = random.Random(n)
  for trial in range(30000):
    if trial % 3 == 0:
      v = [rng.randint(0, 1) for _ in range(n)]
    elif trial % 3 == 1:
      v = [rng.random() for _ in range(n)]
    else:
      v = [rng.randint(0, 5) for _ in range(n)]
    res = run(net, v)
    assert [res[w] for w in out] == sorted(v), n


def main():
  out = sys.stdout
  out.write('// GENERATED by gen_sortnet3.py -- do not edit.  Merge-sort networks of compare-exchanges and 3-sorters.\n')
  out.write('#pragma once\n\nnamespace wbx {\n\n')
  out.write('// sort(x, ...): ascending, x[N] in registers; mn / mx: two-input min / max, mn3 / md3 / mx3: three-input min /\n'
            '// median / max (v_min3_f32, v_med3_f32, v_max3_f32).  NCX / NS3: compare-exchanges / 3-sorters; NINSTR = 2 NCX + 3 NS3.\n')
  out.write('template <int N> struct SortNet3;\n\n')
  out.write('#define WBX_CX(i, j) { const T lo_ = mn(x[i], x[j]); const T hi_ = mx(x[i], x[j]); x[i] = lo_; x[j] = hi_; }\n')
  out.write('#define WBX_S3(i, j, k) { const T lo_ = mn3(x[i], x[j], x[k]); const T md_ = md3(x[i], x[j], x[k]); '
            'const T hi_ = mx3(x[i], x[j], x[k]); x[i] = lo_; x[j] = md_; x[k] = hi_; }\n\n')
  for n in SIZES:
    net, order = build(list(range(n)))
    verify(n, net, order)
    ncx = sum(op[0] == 'ce' for op in net)
    ns3 = len(net) - ncx
    out.write(f'// n = {n}: {ncx} compare-exchanges + {ns3} 3-sorters = {2 * ncx + 3 * ns3} instructions\n')
    out.write(f'template <> struct SortNet3<{n}> {{\n')
    out.write(f'  static constexpr int NCX = {ncx}, NS3 = {ns3}, NINSTR = {2 * ncx + 3 * ns3};\n')
    out.write('  template <typename T, typename MN, typename MX, typename MN3, typename MD3, typename MX3>\n')
    out.write(f'  __device__ __forceinline__ static void sort(T (&x)[{n}], MN mn, MX mx, MN3 mn3, MD3 md3, MX3 mx3) {{\n')
    line = '   '
    for op in net:
      item = f' WBX_CX({op[1]}, {op[2]})' if op[0] == 'ce' else f' WBX_S3({op[1]}, {op[2]}, {op[3]})'
      if len(line) + len(item) > 118:
        out.write(line + '\n')
        line = '   '
      line += item
    out.write(line + '\n')
    out.write(f'    // the sorted order is a wire order (registers are renamed, nothing moves)\n')
    out.write(f'    const T y[{n}] = {{' + ', '.join(f'x[{w}]' for w in order) + '};\n')
    out.write('#pragma unroll\n')
    out.write(f'    for (int i = 0; i < {n}; ++i) x[i] = y[i];\n')
    out.write('  }\n};\n\n')
  out.write('#undef WBX_CX\n#undef WBX_S3\n\n}  // namespace wbx\n')
  sys.stderr.write(', '.join(f'n = {n}: {plan(n)[0]} instructions' for n in SIZES) + '\n')


if __name__ == '__main__':
  main()
