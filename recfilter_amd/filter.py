"""Host-side mirror of the reference's RecFilter front-end (lib/recfilter.h:146-510).

Same names, argument meaning and misuse behaviour as the reference; the two unavoidable
differences of a Halide-free runtime:
  * the defining right-hand side is a bound input -- one device tensor per Tuple element --
    instead of a Halide::Expr:      R.define([x, y], [image])   or   R[x, y] = image
  * realize() returns the output device tensors instead of a Halide::Realization.
Where the reference prints to cerr and assert(false)s, this raises RecFilterUsageError.

Everything numerical happens behind the C ABI (recfilter_amd/plan.py -> librecfilter_amd.so).
"""
from __future__ import annotations

import itertools
from typing import Dict, List, Optional, Sequence, Union

from . import capi
from .capi import RecFilterError
from .plan import Plan

_counter = itertools.count()


class RecFilterUsageError(RuntimeError):
    """The misuse cases the reference reports on cerr before assert(false)."""


class RecFilterDim:
    """lib/recfilter.h:68-95 -- filter dimension: a name and the image extent along it."""

    def __init__(self, var_name: str, var_extent: int):
        self._name, self._extent = str(var_name), int(var_extent)

    def var(self) -> str:
        return self._name

    def num_pixels(self) -> int:
        return self._extent

    def __pos__(self):   # +x : causal scan (lib/recfilter.h:135)
        return RecFilterDimAndCausality(self, True)

    def __neg__(self):   # -x : anticausal scan (lib/recfilter.h:139)
        return RecFilterDimAndCausality(self, False)

    def __repr__(self):
        return f"RecFilterDim({self._name!r}, {self._extent})"


class RecFilterDimAndCausality:
    """lib/recfilter.h:98-128."""

    def __init__(self, rec_var: RecFilterDim, causal: bool):
        self._r, self._c = rec_var, bool(causal)

    def var(self) -> str:
        return self._r.var()

    def num_pixels(self) -> int:
        return self._r.num_pixels()

    def causal(self) -> bool:
        return self._c


class RecFilterSchedule:
    """lib/recfilter.h:516-566.  The reference's schedule handles steer Halide's code generator;
    here the kernels are hand-written, so every directive is accepted and recorded only."""

    def __init__(self, owner: "RecFilter", what: str):
        self._owner, self._what = owner, what

    def _note(self, name, *args):
        self._owner._contents["schedule_log"].append((self._what, name, args))
        return self

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return lambda *args: self._note(name, *args)


class Pointwise:
    """A pointwise consumer of a filter, `w_filtered * F + w_input * I + bias` with I the filter's own
    (prologue-transformed) input -- e.g. the unsharp mask `(1+w)*I - w*Blur` of
    apps/usm/unsharp_mask_optimized.cpp:57.  Passed to RecFilter.compute_at()."""

    def __init__(self, w_filtered: float = 1.0, w_input: float = 0.0, bias: float = 0.0):
        self.w_filtered, self.w_input, self.bias = float(w_filtered), float(w_input), float(bias)


class RecFilter:
    """lib/recfilter.h:146-510.  Copying a RecFilter aliases the same contents
    (lib/recfilter.cpp:141-144); use `RecFilter(other)` for that."""

    _max_threads_per_cuda_warp = 128
    _vectorization_width = 8

    def __init__(self, name: Union[str, "RecFilter"] = ""):
        if isinstance(name, RecFilter):
            self._contents = name._contents        # handle semantics
            return
        self._contents: Dict = dict(
            name=(name or "R") + f"_{next(_counter)}", dims=[], inputs=None, scans=[],
            clamped=False, tiled=False, tile={}, compiled=False, plan=None, schedule_log=[],
            source=None, prologue=None, epilogue=None)

    # -- definition -----------------------------------------------------------------------------
    def name(self) -> str:
        return self._contents["name"]

    def set_clamped_image_border(self) -> None:
        # lib/recfilter.cpp:252-258: must precede the definition
        if self._contents["dims"]:
            raise RecFilterUsageError(f"Recursive filter {self.name()} already defined")
        self._contents["clamped"] = True

    def define(self, pure_args: Sequence[RecFilterDim], pure_def, scale: float = 1.0, bias: float = 0.0) -> None:
        """lib/recfilter.cpp:192-248.  pure_def: one device tensor per Tuple element, or a RecFilter
        whose result feeds this one (the `f2(x,y) = f1.as_func()(x,y)` idiom).  `scale`/`bias` stand for an
        affine defining expression such as `in(x,y)/255` (demo/demo_gaussian_filter.cpp:51-53); they are
        applied on the fly when the passes load pixels."""
        c = self._contents
        if c["dims"]:
            raise RecFilterUsageError(f"Recursive filter {self.name()} already defined")
        if not pure_args:
            raise RecFilterUsageError("a filter needs at least one dimension")
        if isinstance(pure_def, RecFilter):
            c["source"] = pure_def
            inputs = None
        else:
            inputs = list(pure_def) if isinstance(pure_def, (list, tuple)) else [pure_def]
            if not inputs:
                raise RecFilterUsageError("empty definition")
            if any(t.dtype != inputs[0].dtype for t in inputs):
                # lib/recfilter.cpp:198-203
                raise RecFilterUsageError("Type of all Tuple elements in filter definition must be same")
            want = tuple(d.num_pixels() for d in reversed(list(pure_args)))
            for t in inputs:
                if tuple(t.shape) != want:
                    raise RecFilterUsageError(f"input shape {tuple(t.shape)} does not match dimensions {want}")
        c["dims"] = list(pure_args)
        c["inputs"] = inputs
        # only a definition that passed every check above may set the prologue (a rejected re-definition must not
        # change the filter it was rejected for)
        if float(scale) != 1.0 or float(bias) != 0.0:
            c["prologue"] = (float(scale), float(bias))

    def __setitem__(self, dims, value):      # R[x, y] = image
        dims = dims if isinstance(dims, tuple) else (dims,)
        self.define(list(dims), value)

    def add_filter(self, x: Union[RecFilterDim, RecFilterDimAndCausality], coeff: Sequence[float]) -> None:
        """lib/recfilter.cpp:260-392.  coeff = {feedforward, feedback_1 .. feedback_k}."""
        c = self._contents
        if isinstance(x, RecFilterDim):
            x = RecFilterDimAndCausality(x, True)          # lib/recfilter.cpp:260-262
        if not c["dims"]:
            raise RecFilterUsageError(f"Cannot add scans to recursive filter {self.name()} before "
                                      "specifying an initial definition using RecFilter::define()")
        if len(coeff) < 2:
            raise RecFilterUsageError(f"Cannot add scan to recursive filter {self.name()} without "
                                      "feed forward and feedback coefficients")
        names = [d.var() for d in c["dims"]]
        if x.var() not in names:
            raise RecFilterUsageError(f"Variable {x.var()} is not one of the dimensions of the "
                                      f"recursive filter {self.name()}")
        if c["compiled"]:
            raise RecFilterUsageError("cannot add scans after the filter is compiled")
        c["scans"].append((names.index(x.var()), x.causal(), [float(v) for v in coeff]))

    # -- tiling (lib/split.cpp:1850-2080) ---------------------------------------------------------
    def split(self, *args) -> None:
        """split(x, tx[, y, ty[, z, tz]]) or split({name: tile})."""
        c = self._contents
        if len(args) == 1 and isinstance(args[0], dict):
            dims = {str(k): int(v) for k, v in args[0].items()}
        else:
            if len(args) % 2:
                raise RecFilterUsageError("split expects (dim, tile) pairs")
            dims = {args[i].var(): int(args[i + 1]) for i in range(0, len(args), 2)}
        if c["tiled"]:
            raise RecFilterUsageError("Recursive filter cannot be tiled twice")        # split.cpp:1851-1854
        names = [d.var() for d in c["dims"]]
        for var, t in dims.items():
            if var not in names:
                raise RecFilterUsageError(f"Variable {var} is not a dimension of {self.name()}")
            idx = names.index(var)
            if not any(s[0] == idx for s in c["scans"]):
                # split.cpp:1879-1883
                raise RecFilterUsageError(f"Cannot tile dimension {var} without any scans in it")
            if t <= 0 or c["dims"][idx].num_pixels() % t:
                raise RecFilterUsageError(f"tile {t} does not divide the extent of {var}")   # recfilter.h:311
        c["tile"] = dims
        c["tiled"] = True

    def split_all_dimensions(self, tx: int) -> None:
        c = self._contents
        dims = {d.var(): int(tx) for i, d in enumerate(c["dims"]) if any(s[0] == i for s in c["scans"])}
        self.split(dims)

    # -- cascading (lib/reorder.cpp:28-229) -------------------------------------------------------
    def cascade(self, *lists) -> List["RecFilter"]:
        c = self._contents
        if c["tiled"] or c["compiled"]:
            raise RecFilterUsageError("Cascading directive cascade() cannot be used after the filter "
                                      "is already tiled, compiled or realized")
        if len(lists) == 1 and lists[0] and isinstance(lists[0][0], (list, tuple)):
            lists = tuple(lists[0])
        flat = [s for group in lists for s in group]
        n = len(c["scans"])
        for s in flat:
            if not 0 <= s < n:
                raise RecFilterUsageError(f"Scan {s} not found in recursive filter")
        for u, a in enumerate(flat):
            for b in flat[u + 1:]:
                da, ca, _ = c["scans"][a]
                db, cb, _ = c["scans"][b]
                if da == db and ca != cb and b < a:
                    raise RecFilterUsageError(f"Scans {a} {b} cannot be reordered during cascading "
                                              "because they have opposite causality")
        for s in range(n):
            if flat.count(s) == 0:
                raise RecFilterUsageError(f"Scan {s} does not appear in the list of scans for cascading")
            if flat.count(s) > 1:
                raise RecFilterUsageError(f"Scan {s} appears multiple times in the list of scans for cascading")
        out: List[RecFilter] = []
        for i, group in enumerate(lists):
            rf = RecFilter(f"{c['name']}_{i}")
            if c["clamped"]:
                rf.set_clamped_image_border()
            rf.define(c["dims"], (c["inputs"] if c["source"] is None else c["source"]) if i == 0 else out[i - 1])
            if i == 0:
                # the affine defining expression (`in/255`) belongs to the image, so it stays with the stage that
                # reads the image (lib/reorder.cpp:118-133: stage 0 keeps the original pure definition)
                rf._contents["prologue"] = c["prologue"]
            for s in group:
                rf._contents["scans"].append(c["scans"][s])
            out.append(rf)
        return out

    def cascade_by_causality(self) -> List["RecFilter"]:
        scans = self._contents["scans"]
        nd = len(self._contents["dims"])
        causal = [i for d in range(nd) for i, s in enumerate(scans) if s[0] == d and s[1]]
        anti = [i for d in range(nd) for i, s in enumerate(scans) if s[0] == d and not s[1]]
        return self.cascade([causal, anti])

    def cascade_by_dimension(self) -> List["RecFilter"]:
        scans = self._contents["scans"]
        nd = len(self._contents["dims"])
        groups = [[i for i, s in enumerate(scans) if s[0] == d] for d in range(nd)]
        return self.cascade([g for g in groups if g])

    def overlap_to_higher_order_filter(self, fA: "RecFilter", name: str = "O") -> "RecFilter":
        """lib/reorder.cpp:231-381: this filter reads fA's result; merge both into one filter whose
        i-th scan of each dimension has the product transfer function (overlap_feedback_coeff)."""
        from .plan import overlap_feedback_coeff
        a, b = fA._contents, self._contents
        if a["tiled"] or b["tiled"]:
            raise RecFilterUsageError("overlap_to_higher_order_filter cannot be used on tiled filters")
        if len(a["dims"]) != len(b["dims"]):
            raise RecFilterUsageError("filters must have the same dimensions")
        rf = RecFilter(name)
        if a["clamped"]:
            rf.set_clamped_image_border()
        rf.define(a["dims"], a["inputs"] if a["source"] is None else a["source"])
        rf._contents["prologue"] = a["prologue"]       # fA's defining expression is the merged filter's (reorder.cpp:231-381)
        for d in range(len(a["dims"])):
            sa = [s for s in a["scans"] if s[0] == d]
            sb = [s for s in b["scans"] if s[0] == d]
            if len(sa) != len(sb) or any(x[1] != y[1] for x, y in zip(sa, sb)):
                raise RecFilterUsageError("each scan of each dimension must have the same causality in both filters")
            for x, y in zip(sa, sb):
                fb = overlap_feedback_coeff(x[2][1:], y[2][1:])
                rf._contents["scans"].append((d, x[1], [x[2][0] * y[2][0]] + fb))
        return rf

    def compute_at(self, consumer: Pointwise) -> None:
        """lib/recfilter.cpp:473-573: compute this filter's result inside a consumer's tiles instead of writing it to
        memory first.  The consumer is a Pointwise combination of the result and the filter's input; the final pass
        applies it to every sample before the only store (no extra pass over the image)."""
        c = self._contents
        if not isinstance(consumer, Pointwise):
            raise RecFilterUsageError("compute_at takes a Pointwise consumer")
        if c["epilogue"] is not None:
            raise RecFilterUsageError(f"Cannot compute {self.name()} at another consumer because it already has a consumer")
        if c["compiled"]:
            raise RecFilterUsageError("compute_at must be called before the filter is compiled or realized")
        c["epilogue"] = (consumer.w_filtered, consumer.w_input, consumer.bias)

    # -- schedules: accepted, recorded, not needed (lib/recfilter.cpp:396-870) -------------------
    def intra_schedule(self, id: int = 0) -> RecFilterSchedule:
        return RecFilterSchedule(self, f"intra{id}")

    def inter_schedule(self) -> RecFilterSchedule:
        return RecFilterSchedule(self, "inter")

    def full_schedule(self) -> RecFilterSchedule:
        if self._contents["tiled"]:
            raise RecFilterUsageError("Filter is tiled, use RecFilter::intra_schedule() and RecFilter::inter_schedule()")
        return RecFilterSchedule(self, "full")

    def gpu_auto_schedule(self, tile_width: int = 32) -> None: ...
    def gpu_auto_full_schedule(self, tile_width: int = 32) -> None: ...
    def gpu_auto_inter_schedule(self) -> None: ...
    def gpu_auto_intra_schedule(self, id: int = 0) -> None: ...
    def cpu_auto_schedule(self) -> None: ...
    def cpu_auto_full_schedule(self) -> None: ...
    def cpu_auto_inter_schedule(self) -> None: ...
    def cpu_auto_intra_schedule(self) -> None: ...

    @classmethod
    def set_max_threads_per_cuda_warp(cls, v: int) -> None:
        if v % 32:
            raise RecFilterUsageError("max threads per warp must be a multiple of 32")    # recfilter.cpp:39-47
        cls._max_threads_per_cuda_warp = int(v)

    @classmethod
    def set_vectorization_width(cls, v: int) -> None:
        if v not in (2, 4, 8, 16, 32, 64):
            raise RecFilterUsageError("vectorization width must be a power of two <= 64")  # recfilter.cpp:49-57
        cls._vectorization_width = int(v)

    # -- compile and run (lib/recfilter.cpp:918-1016) -------------------------------------------
    def _root_inputs(self):
        """The tensors at the head of the cascade this filter belongs to.  A stage's planes have the shape, device and
        plane count of the head's inputs and the head's PIXEL type (float32 when the head reads unsigned bytes), so a
        plan can be built without launching the upstream stages."""
        c = self._contents
        while c["source"] is not None:
            c = c["source"]._contents
        return c["inputs"]

    # Set False to run the stages of a cascade as separate plans, each reading the previous stage's device buffer (the
    # reference's structure; 12 bytes per sample and stage boundary more).
    merge_cascades = True

    def _cascade_chain(self):
        """The stages of the cascade this filter ends, head first, when the whole chain IS one filter: the scans of a
        cascade are the scans of the filter it was made from (lib/reorder.cpp:100-176 distributes them over Funcs that read
        one another), scans of different dimensions commute and the scans of a dimension stay in order -- so the last
        stage's result is the result of ONE plan on the head's input with all the scans in stage order.  That plan moves
        every sample once per pass instead of once per pass AND stage: gaussian_1xy_2xy (apps/gaussian/
        gaussian_filter_1xy_2xy.cpp:44-54) is four scans per dimension of order <= 2, i.e. one fused stage.  None when a
        stage boundary carries something the merged plan cannot express (a consumer fused into an upstream stage, a
        defining expression on a downstream one, stages of different borders or extents)."""
        c = self._contents
        if not RecFilter.merge_cascades or c["source"] is None or not isinstance(c["source"], RecFilter):
            return None
        chain = [self]
        while isinstance(chain[0]._contents["source"], RecFilter):
            chain.insert(0, chain[0]._contents["source"])
        if chain[0]._contents["source"] is not None:
            return None
        sig = [(d.var(), d.num_pixels()) for d in c["dims"]]
        for st in chain:
            sc = st._contents
            if [(d.var(), d.num_pixels()) for d in sc["dims"]] != sig or sc["clamped"] != c["clamped"]:
                return None
            if sc["tile"] != c["tile"]:          # (stages split to different tile widths stay plans of their own)
                return None
        if any(st._contents["epilogue"] is not None for st in chain[:-1]) or any(st._contents["prologue"] is not None for st in chain[1:]):
            return None
        return chain

    def compile_jit(self, filename: str = "", path: Optional[int] = None) -> None:
        c = self._contents
        if not c["dims"]:
            raise RecFilterUsageError("filter has no definition")
        inputs = self._root_inputs()
        shape = tuple(d.num_pixels() for d in reversed(c["dims"]))
        tile = [c["tile"].get(d.var(), 0) for d in c["dims"]]
        chain = self._cascade_chain()
        scans, prologue, tiled = c["scans"], c["prologue"], c["tiled"]
        if chain is not None:
            scans = [s for st in chain for s in st._contents["scans"]]
            prologue = chain[0]._contents["prologue"]
            tiled = any(st._contents["tiled"] for st in chain)
        asked_path = path
        if path is None:
            path = capi.RF_PATH_AUTO if tiled else capi.RF_PATH_UNTILED
        kw = dict(dtype=inputs[0].dtype, clamped=c["clamped"], planes=len(inputs), tile=tile, device=inputs[0].device.index or 0,
                  epilogue=c["epilogue"])
        try:
            c["plan"] = Plan(shape, scans, path=path, prologue=prologue, **kw)
            c["merged_stages"] = len(chain) if chain is not None else 0
        except (RecFilterError, ValueError):
            # The merged plan of a cascade may not exist where every stage's own plan does -- more than RF_MAX_SCANS scans in
            # all, a combination no path accepts: fall back to the chain of per-stage plans (this stage reads its source's
            # result), as before the merge existed.
            if chain is None:
                raise
            if asked_path is None:
                path = capi.RF_PATH_AUTO if c["tiled"] else capi.RF_PATH_UNTILED
            c["plan"] = Plan(shape, c["scans"], path=path, prologue=c["prologue"], **kw)
            c["merged_stages"] = 0
        c["merge_cascades_at_compile"] = RecFilter.merge_cascades
        c["compiled"] = True

    def _execute_chain(self, fresh_outputs: bool):
        """Launches every upstream cascade stage, then this one (asynchronous).  Like Func::realize on the last stage of
        a cascade, which recomputes all of its producers (they are compute_root Funcs)."""
        c = self._contents
        if c["compiled"] and c.get("merge_cascades_at_compile") != RecFilter.merge_cascades:
            c["compiled"] = False               # merge_cascades was toggled since: the plan no longer says what it should run
        if not c["compiled"]:
            self.compile_jit()
        if c.get("merged_stages"):
            inputs = self._root_inputs()          # the whole cascade is this one plan (compile_jit, _cascade_chain)
        else:
            inputs = c["source"]._execute_chain(False) if c["source"] is not None else c["inputs"]
        reuse = None if fresh_outputs else c.get("outputs")
        c["outputs"] = c["plan"].execute(inputs, reuse)
        return c["outputs"]

    def realize(self):
        """Compute the filter; returns the list of output device tensors (one per Tuple element)."""
        return self._execute_chain(True)

    def profile(self, iterations: int) -> float:
        """lib/recfilter.cpp:991-1016: one warm-up, then the mean wall time of `iterations` runs (ms), every run
        including the upstream stages of a cascade.  Unlike the reference this synchronises the device before
        reading the clock."""
        import time
        import torch
        self._execute_chain(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(int(iterations)):
            self._execute_chain(False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1000.0 / max(int(iterations), 1)

    def plan(self) -> Plan:
        if not self._contents["compiled"]:
            self.compile_jit()
        return self._contents["plan"]

    def print_synopsis(self) -> str:
        c = self._contents
        lines = [f"RecFilter {c['name']}: dims " + ", ".join(f"{d.var()}={d.num_pixels()}" for d in c["dims"])]
        for i, (dim, causal, coeff) in enumerate(c["scans"]):
            lines.append(f"  scan {i}: {'+' if causal else '-'}{c['dims'][dim].var()} {coeff}")
        if c["plan"] is not None:
            lines.append(f"  plan: path={c['plan'].path_name} tiles={c['plan'].tiles}")
        return "\n".join(lines)

    __str__ = print_synopsis
