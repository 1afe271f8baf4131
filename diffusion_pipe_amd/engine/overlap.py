"""Backward progress marks: which gradients of a stage are FINAL while its last backward of a step is still running.

The reference reduces gradients over the data-parallel group after the last backward of a step has finished (DeepSpeed's ReduceGrads behind
utils/patches.py:153-156, the optimizer step of train.py:843-844); SURVEY.md section 8(e) asks for the all-reduce UNDER the tail of that backward.
A stage is a sequence of layers (`module.forward_funcs`), and the backward walks them last to first: once the backward has entered layer j - 1, every
gradient of the layers >= j is final for this micro-batch.  That moment is observable from inside autograd: a tensor hook on a NON-LEAF input of layer j
runs right before the node that produced the tensor executes -- a node created before layer j's forward began -- and the engine's ready queue is a
priority queue on the creation order, so every node created later (all of layers >= j, their weight-gradient kernels and AccumulateGrad nodes
included) has executed by then.  Leaf inputs are never used: their AccumulateGrad runs at top priority, i.e. possibly before the rest of layer j.

`BackwardMarks` installs one forward pre-hook per chosen layer boundary; while a `sink` is set, the forward registers the tensor hooks and the
backward calls `sink(j)` ONCE per boundary, in descending j.  Three sinks exist (engine.py):
  * eager path (CPU / gloo, GPU without graphs): starts the asynchronous average of the gradients of layers >= j;
  * hipGraph lanes / stage graphs: `dpipe_mark_post` -- a one-thread KERNEL NODE of the captured graph that writes the replay's number into the mark's
    device word (include/dpipe_hip.h C5); after the step's last replays were launched the communication stream runs `dpipe_mark_wait` on it.
Boundaries are chosen by gradient bytes: at most `max_marks` of them, none before `min_bytes` of gradients are behind it."""
import torch
from torch import nn


def _layer_modules(module):
    return [f for f in module.forward_funcs]


class BackwardMarks:
    def __init__(self, module, max_marks=6, min_bytes=1 << 20):
        self.layers = _layer_modules(module)
        self.layer_of = {}                      # id(param) -> local layer index (first layer that owns it)
        self.bytes_of_layer = [0] * len(self.layers)
        for j, layer in enumerate(self.layers):
            if not isinstance(layer, nn.Module):
                continue
            for p in layer.parameters():
                if p.requires_grad and id(p) not in self.layer_of:
                    self.layer_of[id(p)] = j
                    self.bytes_of_layer[j] += p.numel() * p.element_size()
        self.boundaries = self._choose(max_marks, min_bytes)
        self.sink = None                        # callable(j) while the forward / backward being run should report its marks
        self._handles = []
        self._generation = 0
        self._fired = set()
        for j in self.boundaries:
            layer = self.layers[j]
            self._handles.append(layer.register_forward_pre_hook(self._make_pre_hook(j)))

    def _choose(self, max_marks, min_bytes):
        """Layer indices j >= 1 (modules with a hookable forward) such that the gradients behind each mark (layers >= j, minus those behind the next mark) hold
        about total / (max_marks + 1) bytes; the gradients in front of the first mark are reduced when the backward has finished, as before."""
        total = sum(self.bytes_of_layer)
        if total == 0 or max_marks <= 0:
            return []
        want = max(min_bytes, total // (max_marks + 1))
        out, acc = [], 0
        for j in range(len(self.layers) - 1, 0, -1):
            acc += self.bytes_of_layer[j]
            if acc >= want and isinstance(self.layers[j], nn.Module) and sum(self.bytes_of_layer[:j]) > 0:
                out.append(j)
                acc = 0
                if len(out) == max_marks:
                    break
        return sorted(out)

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def begin(self):
        """A new forward whose backward should report: marks of earlier forwards go stale."""
        self._generation += 1
        self._fired = set()

    def fired(self):
        """The boundaries whose mark the backward since the last begin() has reported (under a graph capture: the marks that ARE nodes of that graph -- a boundary
        whose layer received no non-leaf float input has none)."""
        return tuple(sorted(self._fired))

    def _make_pre_hook(self, j):
        def pre_hook(_layer, args):
            sink = self.sink
            if sink is None or not torch.is_grad_enabled():
                return None
            gen = self._generation
            tensors = []
            stack = list(args)
            while stack:
                a = stack.pop()
                if torch.is_tensor(a):
                    if a.is_floating_point() and a.requires_grad and a.grad_fn is not None:
                        tensors.append(a)
                elif isinstance(a, (tuple, list)):
                    stack.extend(a)

            def fire(_grad):
                # the first of this boundary's tensors whose producer is about to run: every node of the layers >= j has executed
                if self.sink is sink and self._generation == gen and j not in self._fired:
                    self._fired.add(j)
                    sink(j)
                return None
            for t in tensors:
                t.register_hook(fire)
            return None
        return pre_hook

    # ------------------------------------------------------------------------------------------------ arena geometry
    def arena_bounds(self, params, arenas, grad_of=None):
        """{j: {dtype: element offset}}: the gradients of the layers >= j are exactly arena[dtype][offset:] -- or no entry for a dtype whose arena is not laid out in
        layer order (then that arena is reduced after the backward, as a whole).  `params`: the parameters whose gradient (`grad_of(p)`, default p.grad) may live
        in `arenas` ({dtype: flat})."""
        spans = {dt: [] for dt in arenas}
        base = {dt: a.untyped_storage().data_ptr() for dt, a in arenas.items()}
        for p in params:
            g = p.grad if grad_of is None else grad_of(p)
            if g is None or base.get(g.dtype) != g.untyped_storage().data_ptr():
                continue
            lo = g.storage_offset()
            hi = lo + 1 + sum((s - 1) * st for s, st in zip(g.shape, g.stride()) if s > 0) if g.numel() else lo
            spans[g.dtype].append((lo, hi, self.layer_of.get(id(p), -1)))
        out = {j: {} for j in self.boundaries}
        for dt, sp in spans.items():
            if not sp:
                continue
            sp.sort()
            ordered = all(a[2] <= b[2] for a, b in zip(sp, sp[1:])) and all(a[1] <= b[0] or a[2] == b[2] for a, b in zip(sp, sp[1:]))
            if not ordered:
                continue
            for j in self.boundaries:
                behind = [lo for lo, _, layer in sp if layer >= j]
                if behind:
                    out[j][dt] = min(behind)
        return out
