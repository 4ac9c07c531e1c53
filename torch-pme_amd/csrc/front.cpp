// Compiled front end of the reference call sequence:
//
//     d = pair_distances(positions, pairs, cell, shifts)            tests/helpers.py:278-304
//     V = calculator(charges, cell, positions, pairs, d)            calculators/calculator.py:103-189
//     E = (charges * V).sum(); E.backward()                         README.rst:112-114, tuning/tuner.py:337-373
//
// The kernels are libmipme's (include/mipme.h); what this file replaces is the HOST side of the two autograd nodes for the
// common case -- single-channel mesh calculator, 1/r or 1/r^6 with a smearing, fully periodic, no masks, gradients wanted for
// the positions and, from round 4, for the charges and the cell as well (the whole contract of
// tests/calculators/test_workflow.py:164-192) -- which in Python costs ~0.3 ms per step at 32 000 atoms against ~0.12 ms of kernels (two
// torch.autograd.Function round trips, a dozen torch.empty, ctypes marshalling; profiles/r03_d_prof_dropin.txt).  Here both
// nodes are C++ autograd nodes, each direction is one function that fills the versioned argument structs and calls the C-ABI,
// and the scratch of a call is two allocations.  Anything outside the common case returns None and the Python path
// (ops.py) runs as before: this file adds no behaviour, only removes host time.
//
// Semantics kept from ops.py:
//   * the distances node carries the provenance (positions, cell, pair topology); the calculator node recognises its own
//     `neighbor_distances` by that node, checks that nothing was modified in between (version counters), and then lets the
//     fused pair kernel recompute the distances from the positions;
//   * the pair part of dE/dpositions reaches `positions` without a (P,) gradient for `neighbor_distances` -- UNLESS somebody
//     looks at that gradient (a hook or retain_grad() on d, d among the `inputs=` of autograd.grad): then it is formed with
//     the stand-alone kernel and flows through the distances node, as in the reference's graph (ops.LazyPairGradient does the
//     same lazily in Python);
//   * energy mode (upstream gradient == gE * charges, decided on the device by mipme_scaled_match + a pinned-word poll) uses
//     the per-atom sums of the forward; any other upstream gradient takes the general adjoint kernels;
//   * charges / cell that require a gradient: the forward's gather tail also leaves dE/dq = 2 V and dE/dcell of the energy
//     mode (mipme.h: out_grad_charges, out_grad_cell), the general adjoint forms them for any other upstream gradient, and
//     mipme_energy_select_contract picks on the device.  The pair part of dE/dcell reaches `cell` from the calculator's node
//     directly, like the pair part of dE/dpositions -- unless dE/d(neighbor_distances) is observed.
//
// Build: torch-pme_amd/csrc/Makefile target `front` (g++; no device code here).  libmipme.so is dlopen'ed at the path the
// Python layer loaded it from, so MIPME_LIB builds are honoured.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>

#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/csrc/autograd/function.h>
#include <torch/csrc/autograd/functions/basic_ops.h>
#include <torch/csrc/autograd/graph_task.h>
#include <torch/csrc/autograd/variable.h>
#include <torch/extension.h>

#include "../../include/mipme.h"

namespace {

using torch::autograd::Node;
using torch::autograd::variable_list;

// ---- libmipme entry points ---------------------------------------------------------------------------------------------------
struct Api {
  void* handle = nullptr;
  decltype(&mipme_last_error) last_error = nullptr;
  decltype(&mipme_version) version = nullptr;
  decltype(&mipme_kspace_forward) kspace_forward = nullptr;
  decltype(&mipme_kspace_backward) kspace_backward = nullptr;
  decltype(&mipme_atom_bins_bytes) atom_bins_bytes = nullptr;
  decltype(&mipme_pair_distance_forward_packed) pair_distance_forward_packed = nullptr;
  decltype(&mipme_pair_distance_backward_rows) pair_distance_backward_rows = nullptr;
  decltype(&mipme_scaled_match) scaled_match = nullptr;
  decltype(&mipme_scaled_match_work) scaled_match_work = nullptr;
  decltype(&mipme_scaled_match_wide) scaled_match_wide = nullptr;
  decltype(&mipme_sr_rows_finalize) sr_rows_finalize = nullptr;
  decltype(&mipme_sr_rows_fused) sr_rows_fused = nullptr;
  decltype(&mipme_rspace_backward) rspace_backward = nullptr;
  decltype(&mipme_set_skip_flag) set_skip_flag = nullptr;
  decltype(&mipme_energy_select_sum) energy_select_sum = nullptr;
  decltype(&mipme_energy_select_contract) energy_select_contract = nullptr;
  decltype(&mipme_cell_tail_work) cell_tail_work = nullptr;
  decltype(&mipme_cellgrad_partials_size) cellgrad_partials_size = nullptr;
  decltype(&mipme_rows_partials_size) rows_partials_size = nullptr;
  decltype(&mipme_rspace_rows) rspace_rows = nullptr;
  decltype(&mipme_rspace_rows_tabulated) rspace_rows_tabulated = nullptr;
};
Api g_api;

template <typename F>
void bind(F& fn, const char* name) {
  fn = reinterpret_cast<F>(dlsym(g_api.handle, name));
  if (!fn) throw std::runtime_error(std::string("libmipme lacks ") + name);
}

void load_library(const std::string& path) {
  if (g_api.handle) return;
  g_api.handle = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (!g_api.handle) throw std::runtime_error(std::string("cannot load ") + path + ": " + dlerror());
  bind(g_api.last_error, "mipme_last_error");
  bind(g_api.version, "mipme_version");
  bind(g_api.kspace_forward, "mipme_kspace_forward");
  bind(g_api.kspace_backward, "mipme_kspace_backward");
  bind(g_api.atom_bins_bytes, "mipme_atom_bins_bytes");
  bind(g_api.pair_distance_forward_packed, "mipme_pair_distance_forward_packed");
  bind(g_api.pair_distance_backward_rows, "mipme_pair_distance_backward_rows");
  bind(g_api.scaled_match, "mipme_scaled_match");
  bind(g_api.scaled_match_work, "mipme_scaled_match_work");
  bind(g_api.scaled_match_wide, "mipme_scaled_match_wide");
  bind(g_api.sr_rows_finalize, "mipme_sr_rows_finalize");
  bind(g_api.sr_rows_fused, "mipme_sr_rows_fused");
  bind(g_api.rspace_backward, "mipme_rspace_backward");
  bind(g_api.set_skip_flag, "mipme_set_skip_flag");
  bind(g_api.energy_select_sum, "mipme_energy_select_sum");
  bind(g_api.energy_select_contract, "mipme_energy_select_contract");
  bind(g_api.cell_tail_work, "mipme_cell_tail_work");
  bind(g_api.cellgrad_partials_size, "mipme_cellgrad_partials_size");
  bind(g_api.rows_partials_size, "mipme_rows_partials_size");
  bind(g_api.rspace_rows, "mipme_rspace_rows");
  bind(g_api.rspace_rows_tabulated, "mipme_rspace_rows_tabulated");
  if (g_api.version() != MIPME_VERSION)
    throw std::runtime_error("libmipme version " + std::to_string(g_api.version()) + " != header " + std::to_string(MIPME_VERSION));
}

void check(int rc, const char* what) {
  if (rc != MIPME_OK) throw std::runtime_error(std::string(what) + ": libmipme error " + std::to_string(rc) + ": " + g_api.last_error());
}

int dtype_code(const at::Tensor& t) { return t.scalar_type() == at::kFloat ? MIPME_F32 : MIPME_F64; }

bool same_tensor(const at::Tensor& a, const at::Tensor& b) { return a.unsafeGetTensorImpl() == b.unsafeGetTensorImpl(); }

inline size_t align256(size_t n) { return (n + 255) & ~size_t(255); }

// ---- what the Python layer prepares once per (pair list, shifts) and per (calculator, cell) -------------------------------------
struct FrontTopo {
  // the caller's neighbor_indices: identity + version are the key; held WEAKLY (the list is 76 MB at cfg3 and the Python-side
  // topology cache must not keep every list it has seen alive)
  c10::weak_intrusive_ptr<c10::TensorImpl> pairs{c10::intrusive_ptr<c10::TensorImpl>()};
  uint32_t pairs_version = 0;
  at::Tensor shifts;                         // (P,3) cell shifts in the working dtype
  at::Tensor pairs32, pair_packed;           // list order: the distance kernel
  at::Tensor row_ptr, entries;               // rows
  at::Tensor ent32;                          // rows with shift codes, 4-byte entries: the co-scheduled pair sum of the forward
  int64_t n_atoms = 0, n_pairs = 0;
  // Two more streams are only read by adjoints that the energy mode never launches on a match -- the general pair adjoint
  // (8-byte entries with shift codes) and the distance adjoint (packed row shifts): made on first use by the Python layer
  // (`lazy(kind) -> tensor`), so that a list that is new every call does not pay 0.2 ms for them (cold_list_ms).
  py::object lazy;
  std::mutex lazy_mutex;
  at::Tensor row_packed_, ent_sh_;
  std::atomic<bool> row_packed_ready_{false}, ent_sh_ready_{false};
  // (lock order: the GIL first, then the mutex -- a thread that already holds the GIL may arrive here through a re-entrant
  // backward pass, and must not meet one that holds the mutex and waits for the GIL)
  const at::Tensor& stream_of(at::Tensor& slot, std::atomic<bool>& ready, const char* kind) {
    if (ready.load(std::memory_order_acquire)) return slot;
    py::gil_scoped_acquire gil;
    std::lock_guard<std::mutex> lock(lazy_mutex);
    if (!ready.load(std::memory_order_relaxed)) {
      slot = lazy(kind).cast<at::Tensor>();
      ready.store(true, std::memory_order_release);
    }
    return slot;
  }
  const at::Tensor& row_packed() { return stream_of(row_packed_, row_packed_ready_, "row_packed"); }
  const at::Tensor& ent_sh() { return stream_of(ent_sh_, ent_sh_ready_, "ent_sh"); }
  ~FrontTopo() {
    if (lazy && Py_IsInitialized()) {
      py::gil_scoped_acquire gil;
      lazy = py::object();
    } else {
      lazy.release();
    }
  }
};

bool is_list_of(const at::Tensor& pairs, const FrontTopo& topo) {
  return !topo.pairs.expired() && topo.pairs._unsafe_get_target() == pairs.unsafeGetTensorImpl();
}

struct FrontCalc {
  mipme_mesh_t mesh;
  mipme_potential_t pot;
  mipme_fft_plan* plan = nullptr;
  at::Tensor G;
  at::Tensor G_deriv;  // derivative table of G (mipme_kfilter_build_deriv); undefined: no cell gradients through this file
  at::Tensor cell;  // the cell tensor the geometry and G belong to (identity + version)
  uint32_t cell_version = 0;
  int full_list = 0;
  void* nan_flag = nullptr;
  int64_t n_half = 0;
  py::object keepalive;  // the Python object that owns the plan
  // (the last reference may be dropped by an autograd graph that is torn down without the GIL)
  ~FrontCalc() {
    if (keepalive && Py_IsInitialized()) {
      py::gil_scoped_acquire gil;
      keepalive = py::object();
    } else {
      keepalive.release();
    }
  }
};

// ops.LazyPairGradient (the Python calculator path's placeholder for dE/d(neighbor_distances)) may arrive at a distances node made
// here, when the calculator call itself was outside this file's case:
// `unwrap(g) -> (grad_positions | None, grad_cell | None, plain tensor | None)`
py::object* g_unwrap = nullptr;  // leaked on purpose (no interpreter at static destruction time)
void set_unwrap(py::object fn) { g_unwrap = new py::object(std::move(fn)); }

// The distance adjoint of a backward pass that is itself recorded (create_graph=True), from the differentiable primitives of
// torch-pme_amd/analytic.py: `fn(grad_d, positions, cell, pairs32, shifts, row_ptr, entries, want_pos, want_cell) -> (gp, gc)`.
// Not set: the same expression from ATen's indexing kernels (index_add_ is a compare-and-swap loop in double precision).
py::object* g_recorded_distance_backward = nullptr;
void set_recorded_distance_backward(py::object fn) { g_recorded_distance_backward = new py::object(std::move(fn)); }

// A backward pass that is itself recorded (create_graph=True): the kernels are first order, so their results leave the node
// behind an Error node carrying ops.SECOND_ORDER_HINT -- differentiating them again raises instead of returning an incomplete
// Hessian (what ops.first_order does for the Python nodes).
std::string g_second_order_hint = "torchpme_amd: the HIP kernels provide first-order gradients only";
void set_second_order_hint(const std::string& msg) { g_second_order_hint = msg; }
void first_order_only(at::Tensor& t, const at::Tensor& input) {
  if (!t.defined()) return;
  // (the edge to the node's input keeps the Error node on the path of autograd.grad(..., inputs=[positions]): without it the
  // engine prunes the node and reports an unused input instead of the hint)
  auto err = std::make_shared<torch::autograd::Error>(g_second_order_hint, torch::autograd::collect_next_edges(input));
  torch::autograd::create_gradient_edge(t, err);
}

// ---- the distances node ------------------------------------------------------------------------------------------------------
struct DistNode : public Node {
  std::shared_ptr<FrontTopo> topo;
  at::Tensor pos, cell;  // detached aliases of the inputs
  at::Tensor pos_in, cell_in;  // the inputs as the caller passed them (with their history), for a recorded backward pass
  const void* pos_impl = nullptr;
  const void* cell_impl = nullptr;
  uint32_t pos_version = 0, cell_version = 0;
  const void* dist_impl = nullptr;  // the output (not owned: it owns this node)

  std::string name() const override { return "MipmePairDistancesBackward"; }

  variable_list apply(variable_list&& grads) override {
    variable_list out(2);
    at::Tensor g = grads[0];
    const bool want_pos = task_should_compute_output(0), want_cell = task_should_compute_output(1);
    if (!g.defined() || (!want_pos && !want_cell)) return out;
    if (g.unsafeGetTensorImpl()->is_python_dispatch() && g_unwrap) {
      py::gil_scoped_acquire gil;
      py::tuple r = (*g_unwrap)(g);
      if (!r[0].is_none() || !r[1].is_none()) {  // the fused kernels of the Python path have applied this node's Jacobian already
        if (!r[0].is_none()) out[0] = r[0].cast<at::Tensor>();
        if (!r[1].is_none()) out[1] = r[1].cast<at::Tensor>();
        return out;
      }
      g = r[2].cast<at::Tensor>();
    }
    TORCH_CHECK(pos._version() == pos_version && cell._version() == cell_version,
                "positions or cell of pair_distances() were modified in place before the backward pass");
    if (at::GradMode::is_enabled()) {
      // create_graph=True: the adjoint as differentiable tensor ops -- exact second order, what the reference's helper gives
      // (tests/helpers.py:278-304) and what ops._PairDistances.backward does
      if (g_recorded_distance_backward) {
        py::gil_scoped_acquire gil;
        py::tuple r = (*g_recorded_distance_backward)(g, pos_in, cell_in, topo->pairs32, topo->shifts, topo->row_ptr, topo->entries,
                                                      want_pos, want_cell);
        if (!r[0].is_none()) out[0] = r[0].cast<at::Tensor>();
        if (!r[1].is_none()) out[1] = r[1].cast<at::Tensor>();
        return out;
      }
      at::Tensor i = topo->pairs32.select(1, 0).to(at::kLong), j = topo->pairs32.select(1, 1).to(at::kLong);
      at::Tensor vec = pos_in.index_select(0, j) - pos_in.index_select(0, i) + topo->shifts.matmul(cell_in);
      at::Tensor gvec = (g / at::linalg_vector_norm(vec, 2, at::IntArrayRef{1})).unsqueeze(1) * vec;
      if (want_pos) out[0] = at::zeros_like(pos_in).index_add(0, j, gvec).index_add(0, i, -gvec);
      if (want_cell) out[1] = topo->shifts.t().matmul(gvec);
      return out;
    }
    c10::hip::HIPGuardMasqueradingAsCUDA guard(pos.device());
    auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(pos.device().index()).stream();
    at::Tensor gc = g.contiguous();
    at::Tensor grad_pos = at::empty_like(pos), grad_cell, partials;
    if (want_cell) {
      grad_cell = at::empty_like(cell);
      partials = at::empty({g_api.rows_partials_size(topo->n_atoms)}, pos.options().dtype(at::kDouble));
    }
    check(g_api.pair_distance_backward_rows(stream, dtype_code(pos), topo->n_atoms, topo->row_ptr.data_ptr(),
                                            topo->entries.data_ptr(), topo->row_packed().data_ptr(), pos.data_ptr(),
                                            cell.data_ptr(), nullptr, gc.data_ptr(), want_cell ? partials.data_ptr() : nullptr,
                                            grad_pos.data_ptr(), want_cell ? grad_cell.data_ptr() : nullptr),
          "pair_distance_backward");
    if (want_pos) out[0] = grad_pos;
    if (want_cell) out[1] = grad_cell;
    return out;
  }

  void release_variables() override {
    pos.reset();
    cell.reset();
    pos_in.reset();
    cell_in.reset();
  }
};

bool eligible_real(const at::Tensor& t) {
  return t.defined() && t.is_cuda() && (t.scalar_type() == at::kFloat || t.scalar_type() == at::kDouble) && t.is_contiguous() &&
         t.layout() == at::kStrided;
}

bool stream_is_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) return true;
  return st != hipStreamCaptureStatusNone;
}

// d = |r_j - r_i + S cell| for a prepared list; None if the call is outside this file's case
std::optional<at::Tensor> pair_distances(const std::shared_ptr<FrontTopo>& topo, const at::Tensor& positions, const at::Tensor& cell,
                                         const at::Tensor& pairs) {
  if (!topo || !eligible_real(positions) || !eligible_real(cell) || positions.dim() != 2 || positions.size(1) != 3 ||
      positions.size(0) != topo->n_atoms || cell.dim() != 2 || cell.size(0) != 3 || cell.size(1) != 3 ||
      cell.scalar_type() != positions.scalar_type() || cell.device() != positions.device() ||
      !is_list_of(pairs, *topo) || pairs._version() != topo->pairs_version || topo->n_pairs == 0 ||
      topo->pair_packed.device() != positions.device())
    return std::nullopt;
  const bool need_grad = at::GradMode::is_enabled() && positions.requires_grad();
  if (!need_grad) return std::nullopt;  // without a node there is no provenance to hand to the calculator: Python path
  c10::hip::HIPGuardMasqueradingAsCUDA guard(positions.device());
  auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(positions.device().index()).stream();
  at::Tensor d = at::empty({topo->n_pairs}, positions.options());
  at::Tensor pos = positions.detach(), cl = cell.detach();
  check(g_api.pair_distance_forward_packed(stream, dtype_code(pos), topo->n_pairs, topo->pairs32.data_ptr(),
                                           topo->pair_packed.data_ptr(), pos.data_ptr(), cl.data_ptr(), d.data_ptr()),
        "pair_distance_forward");
  auto node = std::make_shared<DistNode>();
  node->topo = topo;
  node->pos = pos;
  node->cell = cl;
  node->pos_in = positions;
  node->cell_in = cell;
  node->pos_impl = positions.unsafeGetTensorImpl();
  node->cell_impl = cell.unsafeGetTensorImpl();
  node->pos_version = positions._version();
  node->cell_version = cell._version();
  node->dist_impl = d.unsafeGetTensorImpl();
  node->set_next_edges(torch::autograd::collect_next_edges(positions, cell));
  torch::autograd::create_gradient_edge(d, node);
  return d;
}

// ---- the calculator node -----------------------------------------------------------------------------------------------------
thread_local at::Tensor t_match_flag;  // pinned int32[1] per thread: the verdict of mipme_scaled_match
// How the energy mode is decided: 0 = the verdict polled in pinned memory (set_device_select(0)), 2 = on the device for every
// request (set_device_select(2)), 1 (default) = on the device when only the positions want a gradient, polled when the
// charges or the cell do too: the device-side decision launches the whole general adjoint as kernels that return at once, and
// with the charge / cell adjoints on top that is ~12 launches -- more than the wait they avoid (tools/check_front_contract.py:
// E+F+dq 0.149 polled vs 0.168 ms, E+F+dq+dcell 0.168 vs 0.196 ms; positions only 0.131 vs 0.137 on a fast host, 0.171 vs 0.157
// on a slow one).
int g_device_select = 1;
void set_device_select(int mode) { g_device_select = mode; }

struct CalcNode : public Node {
  std::shared_ptr<FrontCalc> calc;
  std::shared_ptr<FrontTopo> topo;
  std::shared_ptr<Node> dist_node;  // DistNode of the distances (also reachable through next_edge(1); kept typed here)
  at::Tensor q, pos, cell, dist;    // detached aliases of the inputs
  at::Tensor pos_in;                // positions as the caller passed them (see first_order_only)
  at::Tensor q_in, cell_in;         // ... and charges / cell, when they require a gradient
  uint32_t q_version = 0, pos_version = 0, cell_version = 0, dist_version = 0;
  at::Tensor force, field;          // per-atom sums of the forward (pair force sums, mesh force field)
  at::Tensor keep;                  // one slab: phi_mesh | rho_dc | atom bins | records
  size_t off_phi = 0, off_dc = 0, off_bins = 0, off_rec = 0;
  // the rest of the contract, per unit of gE, from the forward's gather tail (charges / cell that require a gradient)
  at::Tensor tail_q;                // (N)  dE/dq = 2 V
  at::Tensor tail_cell;             // (27) dE/dcell: mesh part, pair part, sum
  at::Tensor rho_kept, phi_atoms;   // rfftn(rho) and the per-atom mesh potential: the general adjoint's cell gradient

  std::string name() const override { return "MipmeCalculatorBackward"; }

  char* slab(size_t off) const { return static_cast<char*>(keep.data_ptr()) + off; }

  // does anybody look at dE/d(neighbor_distances) itself?
  bool distances_observed() const {
    Node* dn = dist_node.get();
    if (!dn->tensor_pre_hooks().empty() || !dn->retains_grad_hooks().empty() || !dn->pre_hooks().empty()) return true;
    const auto* info = torch::autograd::get_current_graph_task_exec_info();
    if (info && !info->empty()) {
      auto it = info->find(dn);
      if (it != info->end() && it->second.captures_) return true;
    }
    return false;
  }

  variable_list apply(variable_list&& grads) override {
    variable_list out(4);
    const at::Tensor& g_in = grads[0];
    const bool need_pos = task_should_compute_output(0), need_dist = task_should_compute_output(1);
    const bool need_q = tail_q.defined() && task_should_compute_output(2);
    const bool need_cell = tail_cell.defined() && task_should_compute_output(3);
    if (!g_in.defined() || (!need_pos && !need_dist && !need_q && !need_cell)) return out;
    TORCH_CHECK(q._version() == q_version && pos._version() == pos_version && cell._version() == cell_version &&
                    dist._version() == dist_version,
                "an input of the calculator was modified in place before the backward pass");
    const int64_t N = q.size(0), P = topo->n_pairs;
    const int dt = dtype_code(pos);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(pos.device());
    auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(pos.device().index()).stream();
    at::Tensor g = g_in.contiguous();
    const auto opts = pos.options();
    const bool real_dd = need_dist && distances_observed();
    // the mesh part of the cell gradient contains the mesh forces, the pair part the pair forces: both adjoints run for it
    const bool mesh_pos = need_pos || need_cell;
    // (with an observed dE/dd the pair part of every gradient travels through the distances node)
    const bool pair_pos = !real_dd && (need_pos || need_dist || need_cell);

    at::Tensor grad_pos, grad_dist, grad_q, grad_cell;
    at::Tensor cell_mesh, cell_pair;  // general adjoint: the two parts of dL/dcell
    at::Tensor res = at::empty({2}, opts);  // verdict of mipme_scaled_match: {gE, 1 or 0}
    at::Tensor work, cell_partials, pair_partials, q_records;
    // general upstream gradient, mesh part: second spread, convolution and gradient gather
    auto mesh_adjoint = [&](at::Tensor& out_pos) {
      const int64_t nx = calc->mesh.nx, ny = calc->mesh.ny, nz = calc->mesh.nz;
      const size_t s = dt == MIPME_F32 ? 4 : 8;
      const size_t mesh_bytes = align256(size_t(nx) * ny * nz * s), hat_bytes = align256(size_t(calc->n_half) * 2 * s);
      work = at::empty({int64_t(2 * mesh_bytes + hat_bytes + 256)}, opts.dtype(at::kByte));
      char* w = static_cast<char*>(work.data_ptr());
      if (mesh_pos) out_pos = at::empty_like(pos);
      mipme_kspace_backward_args_t a;
      std::memset(&a, 0, sizeof(a));
      a.size = sizeof(a);
      a.version = MIPME_ARGS_VERSION;
      a.plan = calc->plan;
      a.stream = stream;
      a.dtype = dt;
      a.mesh = &calc->mesh;
      a.pot = &calc->pot;
      a.n_atoms = N;
      a.positions = pos.data_ptr();
      a.charges = q.data_ptr();
      a.grad_out = g.data_ptr();
      a.G = calc->G.data_ptr();
      a.phi_mesh = slab(off_phi);
      a.rho_dc = slab(off_dc);
      a.psi_mesh = w;
      a.chi_mesh = w + mesh_bytes;
      a.hat_work = w + 2 * mesh_bytes;
      a.dc = w + 2 * mesh_bytes + hat_bytes;
      a.grad_positions = mesh_pos ? out_pos.data_ptr() : nullptr;
      a.atom_bins = slab(off_bins);
      if (need_q) {
        grad_q = at::empty_like(q);
        a.grad_charges = grad_q.data_ptr();
      }
      if (need_cell) {
        // the fused convolution contracts psi^ with the rfftn(rho) the forward kept (mipme.h, G_deriv)
        cell_mesh = at::empty({3, 3}, opts);
        cell_partials = at::empty({g_api.cellgrad_partials_size(&calc->mesh, N)}, opts.dtype(at::kDouble));
        a.rho_hat = rho_kept.data_ptr();
        a.phi_atoms = phi_atoms.data_ptr();
        a.partials = cell_partials.data_ptr();
        a.grad_cell = cell_mesh.data_ptr();
        a.G_deriv = calc->G_deriv.data_ptr();
      }
      check(g_api.kspace_backward(&a), "kspace_backward");
    };
    // ... pair part, straight to the positions (and the cell) with the fused adjoint kernel
    auto pair_adjoint = [&](at::Tensor& out_pos) {
      out_pos = at::empty_like(pos);
      if (need_cell) {
        cell_pair = at::empty({3, 3}, opts);
        pair_partials = at::empty({g_api.rows_partials_size(N)}, opts.dtype(at::kDouble));
      }
      check(g_api.sr_rows_fused(stream, dt, N, topo->row_ptr.data_ptr(), topo->ent_sh().data_ptr(), topo->entries.data_ptr(), nullptr,
                                pos.data_ptr(), cell.data_ptr(), q.data_ptr(), nullptr, g.data_ptr(), 0, calc->full_list, &calc->pot,
                                0, 1 /* table codes */, slab(off_rec), 1 /* the forward's records: same positions and charges */, nullptr,
                                out_pos.data_ptr(), need_cell ? pair_partials.data_ptr() : nullptr,
                                need_cell ? cell_pair.data_ptr() : nullptr, nullptr),
            "rspace_backward");
    };
    // ... its charge gradient: the transposed pair sum over the upstream gradient, added to the mesh part's
    auto pair_adjoint_charges = [&]() {
      // (records of its own: the kernel repacks them with the upstream gradient in the charge slot, and the forward's must
      // survive for a second backward pass through a retained graph)
      q_records = at::empty({N, 4}, opts);
      check(g_api.sr_rows_fused(stream, dt, N, topo->row_ptr.data_ptr(), topo->ent_sh().data_ptr(), topo->entries.data_ptr(), nullptr,
                                pos.data_ptr(), cell.data_ptr(), q.data_ptr(), g.data_ptr(), nullptr, 1, calc->full_list, &calc->pot,
                                1 /* accumulate */, 1, q_records.data_ptr(), 0, grad_q.data_ptr(), nullptr, nullptr, nullptr, nullptr),
            "rspace_backward_charges");
    };
    // ... or as a (P,) gradient through the distances node
    auto pair_adjoint_dd = [&](const void* scale) {
      grad_dist = at::empty({P}, opts);
      check(g_api.rspace_backward(stream, dt, MIPME_I32, P, N, 1, topo->pairs32.data_ptr(), dist.data_ptr(), q.data_ptr(), nullptr,
                                  calc->full_list, &calc->pot, g.data_ptr(), scale, grad_dist.data_ptr(), nullptr),
            "rspace_backward");
    };
    // charges / cell: the tail's values on a match, the general adjoint's otherwise (decided by the kernel from the verdict)
    auto select_contract = [&](bool matched_on_host) {
      if (!need_q && !need_cell) return;
      if (need_q && !grad_q.defined()) grad_q = at::empty_like(q);
      if (need_cell) grad_cell = at::empty({3, 3}, opts);
      // (a match known on the host: no general parts exist, and the kernel does not read them)
      const void* cm = !need_cell ? nullptr : matched_on_host ? tail_cell.data_ptr() : cell_mesh.data_ptr();
      const void* cp = !need_cell || matched_on_host || !cell_pair.defined() ? nullptr : cell_pair.data_ptr();
      check(g_api.energy_select_contract(stream, dt, N, res.data_ptr(), need_q ? tail_q.data_ptr() : nullptr,
                                         need_q ? grad_q.data_ptr() : nullptr, need_cell ? tail_cell.data_ptr() : nullptr,
                                         real_dd ? 1 : 0, cm, cp, need_cell ? grad_cell.data_ptr() : nullptr),
            "energy_select_contract");
    };
    const bool capturing = stream_is_capturing(stream);
    // g == gE * charges?  (many blocks beyond 32 768 values: one workgroup needs 87 us for 262 144 of them)
    at::Tensor match_work;
    auto run_match = [&](void* flag) {
      const int64_t nw = g_api.scaled_match_work(N);
      if (nw > 0) match_work = at::empty({nw}, opts.dtype(at::kDouble));
      return g_api.scaled_match_wide(stream, dt, N, g.data_ptr(), q.data_ptr(), res.data_ptr(), flag,
                                     nw > 0 ? match_work.data_ptr() : nullptr);
    };

    if (need_pos && !real_dd && !capturing && (g_device_select == 2 || (g_device_select == 1 && !need_q && !need_cell))) {
      // Energy mode (g == gE * charges: the gradient of (charges * V).sum()) decided ON THE DEVICE and acted on there: the general
      // adjoint is launched with every kernel told to return at once on a match, then one kernel writes gE q_a (f force_a +
      // field_a) from the per-atom sums of the forward in that case, or adds up the general adjoint's two parts otherwise.  The
      // host never waits: polling the verdict instead (below) makes it wait for everything queued before, and the GPU then
      // idles while the host prepares the next step.
      at::Tensor flag = at::empty({1}, opts.dtype(at::kInt));
      check(run_match(flag.data_ptr()), "scaled_match");
      struct SkipGuard {
        ~SkipGuard() { g_api.set_skip_flag(nullptr); }
      };
      at::Tensor pair_pos;
      {
        g_api.set_skip_flag(flag.data_ptr());
        SkipGuard guard_skip;
        mesh_adjoint(grad_pos);
        pair_adjoint(pair_pos);
        if (need_q) pair_adjoint_charges();
      }
      check(g_api.energy_select_sum(stream, dt, N, res.data_ptr(), q.data_ptr(), force.data_ptr(), field.data_ptr(), calc->full_list,
                                    grad_pos.data_ptr(), pair_pos.data_ptr(), grad_pos.data_ptr()),
            "energy_select");
      select_contract(false);
    } else {
      // the verdict polled in pinned memory (when dE/d(neighbor_distances) itself is wanted: rare)
      bool match = false;
      if (!capturing) {
        if (!t_match_flag.defined()) t_match_flag = at::empty({1}, at::TensorOptions().dtype(at::kInt).pinned_memory(true));
        volatile int* flag = static_cast<volatile int*>(t_match_flag.data_ptr());
        *flag = -1;
        check(run_match(const_cast<int*>(flag)), "scaled_match");
        int64_t spins = 0;
        while (*flag == -1) {
          if (++spins > 200000000) {  // never seen; a synchronisation is the fallback
            (void)hipStreamSynchronize(stream);
            break;
          }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        match = *flag == 1;
      } else {
        TORCH_CHECK(!need_q && !need_cell,
                    "charge / cell gradients of an eager calculator call cannot be captured into a HIP graph: use "
                    "torchpme_amd.GraphedEnergyForces(charge_gradient=..., cell_gradient=...)");
      }
      if (match) {
        at::Tensor scale = res.narrow(0, 0, 1);
        if (need_pos) {
          grad_pos = at::empty_like(pos);
          // both parts differentiate the same positions: one kernel, gE q_a (f force_a + field_a); with an observed dE/dd the
          // pair part travels through the distances node instead
          check(g_api.sr_rows_finalize(stream, dt, N, real_dd ? nullptr : force.data_ptr(), field.data_ptr(), q.data_ptr(),
                                       scale.data_ptr(), calc->full_list, nullptr, grad_pos.data_ptr(), nullptr),
                "forces_finalize");
        }
        if (real_dd) pair_adjoint_dd(scale.data_ptr());
        select_contract(true);
      } else {
        if (mesh_pos || need_q) mesh_adjoint(grad_pos);
        if (real_dd) {
          pair_adjoint_dd(nullptr);
        } else if (pair_pos) {
          at::Tensor pair_pos_t;
          pair_adjoint(pair_pos_t);
          if (grad_pos.defined())
            grad_pos.add_(pair_pos_t);
          else
            grad_pos = pair_pos_t;
        }
        if (need_q) pair_adjoint_charges();
        if (need_cell) grad_cell = cell_pair.defined() ? cell_mesh + cell_pair : cell_mesh;
      }
    }
    if (at::GradMode::is_enabled()) {
      first_order_only(grad_pos, pos_in);
      first_order_only(grad_dist, pos_in);
      first_order_only(grad_q, q_in.defined() ? q_in : pos_in);
      first_order_only(grad_cell, cell_in.defined() ? cell_in : pos_in);
    }
    if (need_pos) out[0] = grad_pos;
    if (grad_dist.defined()) out[1] = grad_dist;
    if (need_q) out[2] = grad_q;
    if (need_cell) out[3] = grad_cell;
    return out;
  }

  void release_variables() override {
    q.reset();
    pos.reset();
    pos_in.reset();
    q_in.reset();
    cell_in.reset();
    cell.reset();
    dist.reset();
    force.reset();
    field.reset();
    keep.reset();
    tail_q.reset();
    tail_cell.reset();
    rho_kept.reset();
    phi_atoms.reset();
  }
};

// V = per-atom potentials for the prepared calculator; None if the call is outside this file's case
std::optional<at::Tensor> calc_forward(const std::shared_ptr<FrontCalc>& calc, const at::Tensor& charges, const at::Tensor& cell,
                                       const at::Tensor& positions, const at::Tensor& pairs, const at::Tensor& dist) {
  if (!calc || !dist.defined()) return std::nullopt;
  auto fn = dist.grad_fn();
  auto dn = std::dynamic_pointer_cast<DistNode>(fn);
  if (!dn) return std::nullopt;
  const auto& topo = dn->topo;
  if (!at::GradMode::is_enabled() || !eligible_real(positions) || !positions.requires_grad() || !eligible_real(charges) ||
      !eligible_real(cell) || !eligible_real(dist) || dist.retains_grad())
    return std::nullopt;
  // charges / cell that require a gradient: served when the gather tail can carry the energy mode's values -- dE/dq = 2 V
  // needs a half list (mipme.h, out_grad_charges), dE/dcell the derivative table of G
  const bool want_q = charges.requires_grad(), want_cell = cell.requires_grad();
  if ((want_q && calc->full_list) || (want_cell && !calc->G_deriv.defined())) return std::nullopt;
  const int64_t N = topo->n_atoms, P = topo->n_pairs;
  if (charges.dim() != 2 || charges.size(0) != N || charges.size(1) != 1 || positions.dim() != 2 || positions.size(0) != N ||
      positions.size(1) != 3 || dist.dim() != 1 || dist.size(0) != P || pairs.dim() != 2 || pairs.size(0) != P ||
      pairs.size(1) != 2)
    return std::nullopt;
  if (charges.scalar_type() != positions.scalar_type() || cell.scalar_type() != positions.scalar_type() ||
      dist.scalar_type() != positions.scalar_type() || charges.device() != positions.device() ||
      cell.device() != positions.device() || dist.device() != positions.device() || calc->G.device() != positions.device() ||
      calc->G.scalar_type() != positions.scalar_type())
    return std::nullopt;
  // the distances are those of THESE positions, cell and pair list, untouched since
  if (dn->pos_impl != positions.unsafeGetTensorImpl() || dn->pos_version != positions._version() ||
      dn->cell_impl != cell.unsafeGetTensorImpl() || dn->cell_version != cell._version() ||
      dn->dist_impl != dist.unsafeGetTensorImpl() || dist._version() != 0 || !is_list_of(pairs, *topo) ||
      pairs._version() != topo->pairs_version)
    return std::nullopt;
  // ... and the geometry / G(k) are those of this cell
  if (!same_tensor(cell, calc->cell) || cell._version() != calc->cell_version) return std::nullopt;
  if (!topo->ent32.defined()) return std::nullopt;

  c10::hip::HIPGuardMasqueradingAsCUDA guard(positions.device());
  auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(positions.device().index()).stream();
  if (stream_is_capturing(stream)) return std::nullopt;
  const int dt = dtype_code(positions);
  const size_t s = dt == MIPME_F32 ? 4 : 8;
  const int64_t bins_bytes = g_api.atom_bins_bytes(&calc->mesh, N, dt);
  if (bins_bytes <= 0) return std::nullopt;  // meshes the brick kernels do not cover
  const auto opts = positions.options();
  at::Tensor q = charges.detach(), pos = positions.detach(), cl = cell.detach();

  const size_t mesh_bytes = align256(size_t(calc->mesh.nx) * calc->mesh.ny * calc->mesh.nz * s);
  const size_t hat_bytes = align256(size_t(calc->n_half) * 2 * s);
  auto node = std::make_shared<CalcNode>();
  node->off_phi = 0;
  node->off_dc = mesh_bytes;
  node->off_bins = node->off_dc + 256;
  node->off_rec = node->off_bins + align256(size_t(bins_bytes));
  const size_t keep_bytes = node->off_rec + align256(size_t(N) * 4 * s);
  at::Tensor keep = at::empty({int64_t(keep_bytes)}, opts.dtype(at::kByte));
  at::Tensor work = at::empty({int64_t(mesh_bytes + hat_bytes)}, opts.dtype(at::kByte));  // rho_mesh | hat_work
  at::Tensor out = at::empty({N, 1}, opts);
  at::Tensor force = at::empty({N, 3}, opts), field = at::empty({N, 3}, opts);
  at::Tensor tail_e, tail_gp, tail_q, tail_cell, cell_work, rho_kept, phi_atoms;
  if (want_q || want_cell) {
    tail_e = at::empty({}, opts);  // (the tail's energy and assembled forces come with it: unit seed)
    tail_gp = at::empty({N, 3}, opts);
    if (want_q) tail_q = at::empty({N, 1}, opts);
    if (want_cell) {
      tail_cell = at::empty({27}, opts);
      cell_work = at::empty({g_api.cell_tail_work(calc->plan, &calc->mesh, N)}, opts.dtype(at::kDouble));
      rho_kept = at::empty({int64_t(hat_bytes)}, opts.dtype(at::kByte));
      phi_atoms = at::empty({N, 1}, opts);
    }
  }
  char* kp = static_cast<char*>(keep.data_ptr());
  char* wp = static_cast<char*>(work.data_ptr());

  mipme_sr_job_t job;
  std::memset(&job, 0, sizeof(job));
  job.n_atoms = N;
  job.row_ptr = topo->row_ptr.data_ptr();
  job.entries_shift = topo->ent32.data_ptr();
  job.entries = topo->entries.data_ptr();
  job.positions = pos.data_ptr();
  job.cell = cl.data_ptr();
  job.charges = q.data_ptr();
  job.pot = &calc->pot;
  job.full_list = calc->full_list;
  job.shift_format = 2;
  job.records = kp + node->off_rec;
  job.out = out.data_ptr();
  job.force = force.data_ptr();
  job.dist_out = nullptr;

  mipme_kspace_forward_args_t a;
  std::memset(&a, 0, sizeof(a));
  a.size = sizeof(a);
  a.version = MIPME_ARGS_VERSION;
  a.plan = calc->plan;
  a.stream = stream;
  a.dtype = dt;
  a.accumulate_out = 1;
  a.mesh = &calc->mesh;
  a.pot = &calc->pot;
  a.n_atoms = N;
  a.positions = pos.data_ptr();
  a.charges = q.data_ptr();
  a.G = calc->G.data_ptr();
  a.rho_mesh = wp;
  a.rho_hat = nullptr;
  a.hat_work = wp + mesh_bytes;
  a.phi_mesh = kp + node->off_phi;
  a.dc = kp + node->off_dc;
  a.out_lr = out.data_ptr();
  a.atom_bins = kp + node->off_bins;
  a.out_field = field.data_ptr();
  a.out_records = kp + node->off_rec;
  a.sr_job = &job;
  a.nan_flag = calc->nan_flag;
  a.flags = MIPME_FWD_RHO_MESH_UNUSED;  // (rho_mesh is a scratch tensor of this call)
  if (tail_e.defined()) {
    a.out_energy = tail_e.data_ptr();
    a.out_grad_positions = tail_gp.data_ptr();
    if (want_q) a.out_grad_charges = tail_q.data_ptr();
    if (want_cell) {
      a.out_grad_cell = tail_cell.data_ptr();
      a.G_deriv = calc->G_deriv.data_ptr();
      a.cell_work = cell_work.data_ptr();
      a.out_rho_hat = rho_kept.data_ptr();
      a.out_phi = phi_atoms.data_ptr();
    }
  }
  check(g_api.kspace_forward(&a), "kspace_forward");

  node->calc = calc;
  node->topo = topo;
  node->dist_node = fn;
  node->q = q;
  node->pos = pos;
  node->pos_in = positions;
  node->cell = cl;
  node->dist = dist.detach();
  node->q_version = charges._version();
  node->pos_version = positions._version();
  node->cell_version = cell._version();
  node->dist_version = dist._version();
  node->force = force;
  node->field = field;
  node->keep = keep;
  node->tail_q = tail_q;
  node->tail_cell = tail_cell;
  node->rho_kept = rho_kept;
  node->phi_atoms = phi_atoms;
  if (want_q) node->q_in = charges;
  if (want_cell) node->cell_in = cell;
  node->set_next_edges(torch::autograd::collect_next_edges(positions, dist, charges, cell));
  torch::autograd::create_gradient_edge(out, node);
  return out;
}

// ---- the calculator on CALLER-MADE distances -----------------------------------------------------------------------------------
// `neighbor_distances` is a plain tensor without a history (a neighbour-list library's output; the reference tuner's timing
// protocol, tuning/tuner.py:337-373, which clones charges / cell / positions for every call and differentiates result.sum()):
// the pair sum reads the distances as they are -- or the table of v_SR(d) per row entry that ops.PairTopology.tabulated keeps
// for a distance tensor seen before --, gradients go to charges, cell and positions through the mesh part and, for the charges,
// the transposed pair sum.  Always the general adjoint: no energy-mode shortcut here, the Python nodes (~0.2 ms of host time
// per call against ~0.05 here) were the cost of this path, not its kernels.
struct PlainTopo {
  c10::weak_intrusive_ptr<c10::TensorImpl> pairs{c10::intrusive_ptr<c10::TensorImpl>()};
  uint32_t pairs_version = 0;
  at::Tensor row_ptr, entries;
  int64_t n_atoms = 0, n_pairs = 0;
};

struct PlainCalcNode : public Node {
  std::shared_ptr<FrontCalc> calc;
  std::shared_ptr<PlainTopo> topo;
  at::Tensor q, pos, cell, dist;  // detached aliases of the inputs
  at::Tensor q_in, cell_in, pos_in;
  uint32_t q_version = 0, pos_version = 0, cell_version = 0, dist_version = 0;
  at::Tensor keep;  // one slab: phi_mesh | rho_dc | atom bins
  size_t off_phi = 0, off_dc = 0, off_bins = 0;
  at::Tensor rho_kept, phi_atoms;      // cell gradient (see CalcNode)
  at::Tensor tab_values, tab_row_sum;  // v_SR(d) per row entry and its transposed row sums, or undefined

  std::string name() const override { return "MipmeCalculatorPlainDistancesBackward"; }
  char* slab(size_t off) const { return static_cast<char*>(keep.data_ptr()) + off; }

  variable_list apply(variable_list&& grads) override {
    variable_list out(3);
    const at::Tensor& g_in = grads[0];
    const bool need_q = task_should_compute_output(0), need_cell = task_should_compute_output(1),
               need_pos = task_should_compute_output(2);
    if (!g_in.defined() || (!need_q && !need_cell && !need_pos)) return out;
    TORCH_CHECK(q._version() == q_version && pos._version() == pos_version && cell._version() == cell_version &&
                    dist._version() == dist_version,
                "an input of the calculator was modified in place before the backward pass");
    const int64_t N = q.size(0);
    const int dt = dtype_code(pos);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(pos.device());
    auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(pos.device().index()).stream();
    // a uniform upstream gradient (result.sum().backward()): an expanded scalar
    const bool uniform = N > 1 && g_in.dim() == 2 && g_in.stride(0) == 0;
    at::Tensor g = g_in.contiguous();
    const auto opts = pos.options();
    const size_t s = dt == MIPME_F32 ? 4 : 8;
    const size_t mesh_bytes = align256(size_t(calc->mesh.nx) * calc->mesh.ny * calc->mesh.nz * s);
    const size_t hat_bytes = align256(size_t(calc->n_half) * 2 * s);
    at::Tensor work = at::empty({int64_t(2 * mesh_bytes + hat_bytes + 256)}, opts.dtype(at::kByte));
    char* w = static_cast<char*>(work.data_ptr());
    at::Tensor grad_pos, grad_q, grad_cell, cell_partials;
    mipme_kspace_backward_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.size = sizeof(a);
    a.version = MIPME_ARGS_VERSION;
    a.plan = calc->plan;
    a.stream = stream;
    a.dtype = dt;
    a.mesh = &calc->mesh;
    a.pot = &calc->pot;
    a.n_atoms = N;
    a.positions = pos.data_ptr();
    a.charges = q.data_ptr();
    a.grad_out = g.data_ptr();
    a.G = calc->G.data_ptr();
    a.phi_mesh = slab(off_phi);
    a.rho_dc = slab(off_dc);
    a.psi_mesh = w;
    a.chi_mesh = w + mesh_bytes;
    a.hat_work = w + 2 * mesh_bytes;
    a.dc = w + 2 * mesh_bytes + hat_bytes;
    a.atom_bins = slab(off_bins);
    if (need_pos || need_cell) {  // (the cell gradient contains the mesh forces)
      grad_pos = at::empty_like(pos);
      a.grad_positions = grad_pos.data_ptr();
    }
    if (need_q) {
      grad_q = at::empty_like(q);
      a.grad_charges = grad_q.data_ptr();
    }
    if (need_cell) {
      grad_cell = at::empty({3, 3}, opts);
      cell_partials = at::empty({g_api.cellgrad_partials_size(&calc->mesh, N)}, opts.dtype(at::kDouble));
      a.rho_hat = rho_kept.data_ptr();
      a.phi_atoms = phi_atoms.data_ptr();
      a.partials = cell_partials.data_ptr();
      a.grad_cell = grad_cell.data_ptr();
      a.G_deriv = calc->G_deriv.data_ptr();
    }
    check(g_api.kspace_backward(&a), "kspace_backward");
    if (need_q) {
      if (tab_values.defined() && uniform) {
        grad_q.addcmul_(tab_row_sum, g_in.narrow(0, 0, 1));  // c times the table's row sums
      } else if (tab_values.defined()) {
        check(g_api.rspace_rows_tabulated(stream, dt, N, topo->row_ptr.data_ptr(), tab_values.data_ptr(), g.data_ptr(), 1,
                                          calc->full_list, 1, grad_q.data_ptr()),
              "rspace_backward_charges");
      } else {
        check(g_api.rspace_rows(stream, dt, N, 1, topo->row_ptr.data_ptr(), topo->entries.data_ptr(), dist.data_ptr(), g.data_ptr(),
                                nullptr, 1, calc->full_list, &calc->pot, 1, grad_q.data_ptr()),
              "rspace_backward_charges");
      }
    }
    if (at::GradMode::is_enabled()) {
      first_order_only(grad_pos, pos_in.defined() ? pos_in : q_in.defined() ? q_in : cell_in);
      first_order_only(grad_q, q_in.defined() ? q_in : pos_in.defined() ? pos_in : cell_in);
      first_order_only(grad_cell, cell_in.defined() ? cell_in : pos_in.defined() ? pos_in : q_in);
    }
    if (need_q) out[0] = grad_q;
    if (need_cell) out[1] = grad_cell;
    if (need_pos) out[2] = grad_pos;
    return out;
  }

  void release_variables() override {
    q.reset();
    pos.reset();
    cell.reset();
    dist.reset();
    q_in.reset();
    cell_in.reset();
    pos_in.reset();
    keep.reset();
    rho_kept.reset();
    phi_atoms.reset();
    tab_values.reset();
    tab_row_sum.reset();
  }
};

// V for caller-made distances; None if the call is outside this case.  The geometry / G(k) of `calc` are the CALLER's word for
// this cell (the Python layer verifies a speculative geometry after the call, calculators.py).
std::optional<at::Tensor> calc_forward_plain(const std::shared_ptr<FrontCalc>& calc, const std::shared_ptr<PlainTopo>& topo,
                                             const at::Tensor& charges, const at::Tensor& cell, const at::Tensor& positions,
                                             const at::Tensor& pairs, const at::Tensor& dist,
                                             const std::optional<at::Tensor>& tab_values,
                                             const std::optional<at::Tensor>& tab_row_sum) {
  if (!calc || !topo || !at::GradMode::is_enabled() || !eligible_real(positions) || !eligible_real(charges) ||
      !eligible_real(cell) || !eligible_real(dist) || dist.requires_grad() || dist.grad_fn())
    return std::nullopt;
  const bool want_q = charges.requires_grad(), want_cell = cell.requires_grad(), want_pos = positions.requires_grad();
  if ((!want_q && !want_cell && !want_pos) || (want_cell && !calc->G_deriv.defined())) return std::nullopt;
  const int64_t N = topo->n_atoms, P = topo->n_pairs;
  if (N <= 0 || P <= 0 || charges.dim() != 2 || charges.size(0) != N || charges.size(1) != 1 || positions.dim() != 2 ||
      positions.size(0) != N || positions.size(1) != 3 || dist.dim() != 1 || dist.size(0) != P || pairs.dim() != 2 ||
      pairs.size(0) != P || pairs.size(1) != 2 || cell.dim() != 2 || cell.size(0) != 3 || cell.size(1) != 3)
    return std::nullopt;
  if (charges.scalar_type() != positions.scalar_type() || cell.scalar_type() != positions.scalar_type() ||
      dist.scalar_type() != positions.scalar_type() || charges.device() != positions.device() ||
      cell.device() != positions.device() || dist.device() != positions.device() || calc->G.device() != positions.device() ||
      calc->G.scalar_type() != positions.scalar_type() || topo->row_ptr.device() != positions.device())
    return std::nullopt;
  if (topo->pairs.expired() || topo->pairs._unsafe_get_target() != pairs.unsafeGetTensorImpl() ||
      pairs._version() != topo->pairs_version)
    return std::nullopt;
  const bool tab = tab_values.has_value() && tab_row_sum.has_value() && tab_values->defined() && tab_row_sum->defined();

  c10::hip::HIPGuardMasqueradingAsCUDA guard(positions.device());
  auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(positions.device().index()).stream();
  if (stream_is_capturing(stream)) return std::nullopt;
  const int dt = dtype_code(positions);
  const size_t s = dt == MIPME_F32 ? 4 : 8;
  const int64_t bins_bytes = g_api.atom_bins_bytes(&calc->mesh, N, dt);
  if (bins_bytes <= 0) return std::nullopt;
  const auto opts = positions.options();
  at::Tensor q = charges.detach(), pos = positions.detach(), cl = cell.detach();

  const size_t mesh_bytes = align256(size_t(calc->mesh.nx) * calc->mesh.ny * calc->mesh.nz * s);
  const size_t hat_bytes = align256(size_t(calc->n_half) * 2 * s);
  auto node = std::make_shared<PlainCalcNode>();
  node->off_phi = 0;
  node->off_dc = mesh_bytes;
  node->off_bins = node->off_dc + 256;
  const size_t keep_bytes = node->off_bins + align256(size_t(bins_bytes));
  at::Tensor keep = at::empty({int64_t(keep_bytes)}, opts.dtype(at::kByte));
  at::Tensor work = at::empty({int64_t(mesh_bytes + hat_bytes)}, opts.dtype(at::kByte));  // rho_mesh | hat_work
  at::Tensor out = at::empty({N, 1}, opts);
  at::Tensor rho_kept, phi_atoms;
  if (want_cell) {
    rho_kept = at::empty({int64_t(hat_bytes)}, opts.dtype(at::kByte));
    phi_atoms = at::empty({N, 1}, opts);
  }
  char* kp = static_cast<char*>(keep.data_ptr());
  char* wp = static_cast<char*>(work.data_ptr());
  // the pair sum first (it overwrites `out`), then the mesh part on top
  if (tab)
    check(g_api.rspace_rows_tabulated(stream, dt, N, topo->row_ptr.data_ptr(), tab_values->data_ptr(), q.data_ptr(), 0,
                                      calc->full_list, 0, out.data_ptr()),
          "rspace_forward");
  else
    check(g_api.rspace_rows(stream, dt, N, 1, topo->row_ptr.data_ptr(), topo->entries.data_ptr(), dist.data_ptr(), q.data_ptr(), nullptr,
                            0, calc->full_list, &calc->pot, 0, out.data_ptr()),
          "rspace_forward");
  mipme_kspace_forward_args_t a;
  std::memset(&a, 0, sizeof(a));
  a.size = sizeof(a);
  a.version = MIPME_ARGS_VERSION;
  a.plan = calc->plan;
  a.stream = stream;
  a.dtype = dt;
  a.accumulate_out = 1;
  a.mesh = &calc->mesh;
  a.pot = &calc->pot;
  a.n_atoms = N;
  a.positions = pos.data_ptr();
  a.charges = q.data_ptr();
  a.G = calc->G.data_ptr();
  a.rho_mesh = wp;
  a.rho_hat = nullptr;
  a.hat_work = wp + mesh_bytes;
  a.phi_mesh = kp + node->off_phi;
  a.dc = kp + node->off_dc;
  a.out_lr = out.data_ptr();
  a.atom_bins = kp + node->off_bins;
  a.nan_flag = calc->nan_flag;
  a.flags = MIPME_FWD_RHO_MESH_UNUSED;  // (rho_mesh is a scratch tensor of this call)
  if (want_cell) {
    a.out_phi = phi_atoms.data_ptr();
    a.out_rho_hat = rho_kept.data_ptr();
  }
  check(g_api.kspace_forward(&a), "kspace_forward");

  node->calc = calc;
  node->topo = topo;
  node->q = q;
  node->pos = pos;
  node->cell = cl;
  node->dist = dist;
  if (want_q) node->q_in = charges;
  if (want_cell) node->cell_in = cell;
  if (want_pos) node->pos_in = positions;
  node->q_version = charges._version();
  node->pos_version = positions._version();
  node->cell_version = cell._version();
  node->dist_version = dist._version();
  node->keep = keep;
  node->rho_kept = rho_kept;
  node->phi_atoms = phi_atoms;
  if (tab) {
    node->tab_values = *tab_values;
    node->tab_row_sum = *tab_row_sum;
  }
  node->set_next_edges(torch::autograd::collect_next_edges(charges, cell, positions));
  torch::autograd::create_gradient_edge(out, node);
  return out;
}

bool is_front_distances(const at::Tensor& d) {
  return d.defined() && std::dynamic_pointer_cast<DistNode>(d.grad_fn()) != nullptr;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "compiled host side of the reference call sequence (see front.cpp)";
  m.def("load_library", &load_library);
  m.def("set_unwrap", &set_unwrap);
  m.def("set_recorded_distance_backward", &set_recorded_distance_backward);
  m.def("set_device_select", &set_device_select);
  m.def("set_second_order_hint", &set_second_order_hint);
  py::class_<FrontTopo, std::shared_ptr<FrontTopo>>(m, "Topology")
      .def(py::init([](at::Tensor pairs, at::Tensor shifts, at::Tensor pairs32, at::Tensor pair_packed, at::Tensor row_ptr,
                       at::Tensor entries, at::Tensor ent32, int64_t n_atoms, py::object lazy) {
        auto t = std::make_shared<FrontTopo>();
        t->pairs = c10::weak_intrusive_ptr<c10::TensorImpl>(pairs.getIntrusivePtr());
        t->pairs_version = pairs._version();
        t->shifts = shifts;
        t->pairs32 = pairs32;
        t->pair_packed = pair_packed;
        t->row_ptr = row_ptr;
        t->entries = entries;
        t->ent32 = ent32;
        t->n_atoms = n_atoms;
        t->n_pairs = pairs.size(0);
        t->lazy = std::move(lazy);
        return t;
      }));
  py::class_<FrontCalc, std::shared_ptr<FrontCalc>>(m, "Calculator")
      .def(py::init([](py::bytes mesh, py::bytes pot, int64_t plan, at::Tensor G, at::Tensor cell, bool full_list, int64_t nan_flag,
                       int64_t n_half, py::object keepalive, std::optional<at::Tensor> G_deriv) {
        auto c = std::make_shared<FrontCalc>();
        const std::string mb = mesh, pb = pot;
        if (mb.size() != sizeof(mipme_mesh_t) || pb.size() != sizeof(mipme_potential_t))
          throw std::runtime_error("descriptor size mismatch between the ctypes mirror and include/mipme.h");
        std::memcpy(&c->mesh, mb.data(), sizeof(mipme_mesh_t));
        std::memcpy(&c->pot, pb.data(), sizeof(mipme_potential_t));
        c->plan = reinterpret_cast<mipme_fft_plan*>(plan);
        c->G = G;
        if (G_deriv) c->G_deriv = *G_deriv;
        c->cell = cell;
        c->cell_version = cell._version();
        c->full_list = full_list ? 1 : 0;
        c->nan_flag = reinterpret_cast<void*>(nan_flag);
        c->n_half = n_half;
        c->keepalive = std::move(keepalive);
        return c;
      }));
  py::class_<PlainTopo, std::shared_ptr<PlainTopo>>(m, "PlainTopology")
      .def(py::init([](at::Tensor pairs, at::Tensor row_ptr, at::Tensor entries, int64_t n_atoms) {
        auto t = std::make_shared<PlainTopo>();
        t->pairs = c10::weak_intrusive_ptr<c10::TensorImpl>(pairs.getIntrusivePtr());
        t->pairs_version = pairs._version();
        t->row_ptr = row_ptr;
        t->entries = entries;
        t->n_atoms = n_atoms;
        t->n_pairs = pairs.size(0);
        return t;
      }));
  m.def("calc_forward_plain", &calc_forward_plain);
  m.def("pair_distances", &pair_distances);
  m.def("calc_forward", &calc_forward);
  m.def("is_front_distances", &is_front_distances);
}
