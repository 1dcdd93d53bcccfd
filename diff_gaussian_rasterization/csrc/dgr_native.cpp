// Host-only C++ half of the drop-in `diff_gaussian_rasterization` package for MI355X: the autograd nodes of GaussianRasterizer and of
// the fused mapping loss, and the workspace / capacity state machine, written against libtorch's autograd C++ API over the torch-free
// C ABI of libsplat_hip.so (include/splat_hip.h).  No kernel lives here -- every launch goes through sgr_* -- and no CPU path either.
//
// Why it exists.  The reference's loop structure stays intact under the drop-in (north star): up to 12 render() calls per mapping
// iteration, their losses summed, ONE loss.backward() (/root/reference/src/mapper.py:426-490,557; the rasterizer is entered at
// /root/reference/thirdparty/gaussian_splatting/gaussian_renderer/__init__.py:130-141).  With the nodes written as Python
// torch.autograd.Function that iteration was pure host time (round 3: 3.36 ms, of which 3.356 ms enqueue): ~55 us of interpreter
// per rasterizer forward, ~33 us per loss, a Python round trip per node in the backward.  Here a forward is ONE call from Python
// into try_rasterize() (settings struct + lease + sgr_forward + header post + node wiring, a few microseconds besides the
// launches) and the backward never re-enters the interpreter.
//
// Same design as the Python implementation it replaces for the common call (diff_gaussian_rasterization/__init__.py keeps every
// other input combination): the renders of one iteration share their parameter tensors, so the first of them routes those through
// an identity node (CollectNode) whose aliases all of them consume; a view's own node (ViewNode) only records the image gradients
// it is handed; the engine runs the collector exactly once, after the last of them, and there ONE sgr_backward_views call
// does the tile backward, the projection backward and the gather of all views.
#include <torch/extension.h>
#include <torch/csrc/autograd/function.h>
#include <torch/csrc/autograd/variable.h>
#include <torch/csrc/autograd/functions/accumulate_grad.h>
#include <torch/csrc/autograd/generated/Functions.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>

#include <atomic>
#include <chrono>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/splat_hip.h"

namespace py = pybind11;
using at::Tensor;
using torch::autograd::variable_list;

namespace dgr {

constexpr int kRing = 64;
constexpr uint32_t kSentinel = 0xFFFFFFFFu;
constexpr int64_t kCapFloor = 1 << 20;
constexpr int64_t kCapFactor = 4;
static const char* kOverflowMsg =
    "a rasterizer forward exceeded the (tile, Gaussian) pair capacity: its image was rendered from truncated tile lists. "
    "No gradient has been produced and the capacity has been raised -- re-run the iteration "
    "(SPLAT_RASTER_SYNC=1 sizes every forward synchronously and never drops pairs)";

// PyTorch-ROCm tensors carry DeviceType::CUDA: the current stream of the device a tensor lives on, as torch sees it
static inline hipStream_t current_stream(int dev) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA((c10::DeviceIndex)dev).stream(); }
static inline c10::Device cuda_device(int dev) { return c10::Device(c10::DeviceType::CUDA, (c10::DeviceIndex)dev); }

static void check(int rc, const char* what) {
  TORCH_CHECK(rc == SGR_OK, what, " failed (", rc, "): ", sgr_last_error());
}

// fp32, contiguous (what the C ABI takes); a no-op for the tensors the mapping loop passes
static inline Tensor f32c(const Tensor& t) {
  if (!t.defined()) return t;
  Tensor r = t.scalar_type() == at::kFloat ? t : t.to(at::kFloat);
  return r.is_contiguous() ? r : r.contiguous();
}
static inline const float* fptr(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

// A leaf the caller only ever reads `.grad` of (screenspace_points, the pose deltas, the exposure parameters of the mapping loop): with
// deferral on, its gradient is written into `.grad` by the node that produces it -- set, or added to an existing one -- instead of
// travelling through an AccumulateGrad node of its own (5 per view, 60 per 12-view backward pass: ~0.15 ms of engine time).
// `loss.backward()` leaves the same `.grad`s; torch.autograd.grad(...) on such a leaf or tensor hooks on it need
// SPLAT_RASTER_DEFER_POSE_GRADS=0.  Leaves with hooks are never deferred.
static inline bool deferable_leaf(const Tensor& t) {
  if (!t.defined() || !t.requires_grad() || t.grad_fn()) return false;
  return torch::autograd::impl::hooks(t).empty() && !torch::autograd::impl::post_acc_grad_hooks(t);
}
static inline void deposit_grad(const Tensor& leaf, const Tensor& value) {
  Tensor& g = const_cast<Tensor&>(leaf).mutable_grad();
  if (!g.defined()) g = value;
  else g.add_(value);
}

// ---- workspace policy: identical to the Python _DeviceState (see the comment block in diff_gaussian_rasterization/__init__.py) ----
struct Pool {
  std::mutex mu;                   // (a lease may be returned from the thread that drops the last reference of a graph)
  std::vector<Tensor> blocks;
};
// returns a saved block to its pool when the node that holds it dies
struct Lease {
  std::shared_ptr<Pool> pool;
  Tensor block;
  ~Lease() {
    if (!pool) return;
    std::lock_guard<std::mutex> lk(pool->mu);
    if (pool->blocks.size() < 32) pool->blocks.push_back(std::move(block));
  }
};

struct Batch;
struct ViewRecord;
struct PendingHeader {
  int slot;
  int64_t cap;                          // capacity the forward ran with
  std::weak_ptr<ViewRecord> rec;        // the forward's record, if a backward may follow (it can be re-run when it was truncated)
};
static bool mark_truncated(const std::weak_ptr<ViewRecord>& rec, int64_t need);

struct DeviceState {
  std::mutex mu;
  int dev = 0;
  std::vector<Tensor> scratch;
  size_t scratch_bytes = 0;
  int64_t capacity = kCapFloor;
  std::map<std::tuple<int64_t, int, int, int64_t>, std::pair<size_t, size_t>> sizes;
  std::map<std::tuple<int64_t, int, int, int64_t>, std::shared_ptr<Pool>> pools;
  std::pair<int64_t, const void*> last_map{-1, nullptr};
  int64_t last_pairs = 0;
  int64_t longest_list = 0;
  Tensor ring;               // pinned [kRing, 16] int32: headers of recent forwards
  volatile uint32_t* ring_w = nullptr;
  std::deque<PendingHeader> pending;
  int64_t floor_override = -1;                       // tests: a capacity floor below the real one (-1: none)
  int64_t reruns = 0;                                // truncated forwards re-run inside a backward pass
  int next_slot = 0;
  int64_t overflowed = 0, unreported = 0;
  int64_t lost = 0;                                  // truncated forwards nobody can re-run (no graph / outputs dropped): warned about at the next call
  std::shared_ptr<Batch> batch;                      // the batch forwards currently join
  int64_t forwards = 0, batches = 0, backwards = 0;  // counters for tests / profiling

  void ensure_ring() {
    if (ring.defined()) return;
    ring = at::empty({kRing, 16}, at::TensorOptions().dtype(at::kInt).device(at::kCPU).pinned_memory(true));
    ring_w = reinterpret_cast<volatile uint32_t*>(ring.data_ptr<int32_t>());
  }
  int64_t floor_for(int64_t N, int64_t ntiles) const {
    return floor_override >= 0 ? floor_override : std::max<int64_t>({kCapFloor, 8 * N, 512 * ntiles});
  }
  std::pair<size_t, size_t> bytes_for(int64_t N, int H, int W, int64_t cap) {
    auto key = std::make_tuple(N, H, W, cap);
    auto it = sizes.find(key);
    if (it != sizes.end()) return it->second;
    if (sizes.size() > 256) sizes.clear();
    auto v = std::make_pair(sgr_saved_bytes((int32_t)N, H, W, cap), sgr_scratch_bytes((int32_t)N, H, W, cap));
    sizes[key] = v;
    return v;
  }
  Tensor& scratch_block(size_t k, size_t nbytes) {
    if (nbytes > scratch_bytes) {
      scratch.clear();
      scratch_bytes = nbytes;
    }
    while (scratch.size() <= k)
      scratch.push_back(at::empty({(int64_t)scratch_bytes}, at::TensorOptions().dtype(at::kByte).device(at::kCUDA, dev)));
    return scratch[k];
  }
  std::shared_ptr<Lease> lease(int64_t N, int H, int W, int64_t cap, size_t saved_bytes, int* clean) {
    auto key = std::make_tuple(N, H, W, cap);
    auto it = pools.find(key);
    if (it == pools.end()) {
      if (pools.size() > 8) pools.clear();          // map size / capacity changed a few times: forget the old shapes
      it = pools.emplace(key, std::make_shared<Pool>()).first;
    }
    auto l = std::make_shared<Lease>();
    l->pool = it->second;
    std::lock_guard<std::mutex> lk(l->pool->mu);
    if (!l->pool->blocks.empty()) {
      l->block = std::move(l->pool->blocks.back());
      l->pool->blocks.pop_back();
      *clean = 1;                                    // went through a forward with this layout: counters are clean
    } else {
      l->block = at::empty({(int64_t)saved_bytes}, at::TensorOptions().dtype(at::kByte).device(at::kCUDA, dev));
      *clean = 0;
    }
    return l;
  }
  // folds the pair counts that have arrived into `capacity`; returns the number of forwards that dropped pairs
  int64_t drain(bool wait) {
    int64_t bad = 0;
    while (!pending.empty()) {
      const int slot = pending.front().slot;
      const int64_t cap = pending.front().cap;
      volatile uint32_t* w = ring_w + 16 * slot;
      if (w[15] == kSentinel) {                      // the copy has not landed yet
        if (!wait) break;
        // the copy was enqueued right behind its forward: it lands long before later work on the stream finishes, so poll the
        // pinned word for a while before falling back to a stream synchronisation
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
        while (w[15] == kSentinel && std::chrono::steady_clock::now() < t_end) {
        }
        if (w[15] == kSentinel) C10_HIP_CHECK(hipStreamSynchronize(current_stream(dev)));
      }
      const int64_t R = w[0];
      if (w[1] == 2u) {                              // header.overflow == 2: a 16-bit tile counter saturated
        pending.pop_front();
        TORCH_CHECK(false, "more than 65280 splats on one 8x8 tile: the map has degenerated");
      }
      last_pairs = R;
      longest_list = std::max<int64_t>((int64_t)w[10], (longest_list * 7) / 8);      // (decays when lists shrink)
      if (kCapFactor * R > capacity) capacity = kCapFactor * R;
      if (R > cap) {
        ++bad;
        // a forward whose record is gone (rendered without a graph, or its outputs were dropped before a backward): nothing will
        // re-run it -- its image was consumed as it was and the caller is told at the very next call (try_rasterize)
        if (!mark_truncated(pending.front().rec, R)) ++lost;
      }
      pending.pop_front();
    }
    overflowed += bad;
    unreported += bad;
    return bad;
  }
  // raises if a forward since the last report dropped pairs (checked without waiting unless `wait`)
  void report(bool wait) {
    if (!pending.empty()) drain(wait);
    if (unreported) {
      unreported = 0;
      TORCH_CHECK(false, kOverflowMsg);
    }
  }
  void post(const void* saved, int64_t cap, hipStream_t stream, const std::shared_ptr<ViewRecord>& rec = nullptr) {
    ensure_ring();
    if ((int)pending.size() >= kRing - 1) drain(true);
    const int slot = next_slot;
    next_slot = (slot + 1) % kRing;
    ring_w[16 * slot + 15] = kSentinel;              // (the header's last pad word is 0 on the device)
    check(sgr_header_to_host(saved, (void*)(ring_w + 16 * slot), (void*)stream), "sgr_header_to_host");
    pending.push_back(PendingHeader{slot, cap, rec});
  }
};

static std::mutex g_states_mu;
static std::map<int, std::unique_ptr<DeviceState>> g_states;
static DeviceState& state(int dev) {
  std::lock_guard<std::mutex> lk(g_states_mu);
  auto& p = g_states[dev];
  if (!p) {
    p = std::make_unique<DeviceState>();
    p->dev = dev;
  }
  return *p;
}

// ---- batched backward -------------------------------------------------------------------------------------------------------------
// what the collector needs from one forward to run its backward later
struct ViewRecord {
  SgrSettings settings;
  Tensor keep[5];                  // bg, viewmatrix, projmatrix, projmatrix_raw, campos (the settings struct points into them)
  Tensor radii;
  std::shared_ptr<Lease> lease;
  int64_t cap = 0;
  size_t saved_bytes = 0;
  int H = 0, W = 0;
  Tensor grad_color, grad_depth;
  float* m2d_ptr = nullptr;        // this view's dL/dmeans2D [N,3] (a zero-filled slice the view's node handed to autograd)
  Tensor pose_arena;               // strict pose mode: rho[3] | theta[3] handed to autograd
  Tensor means2D, theta, rho;      // the caller's leaves that want a gradient (else undefined)
  bool armed = false;
  bool strict_pose = false;
  bool m2d_deferred = false;       // means2D is a hook-free leaf and deferral is on: the node writes `.grad` itself
  int64_t truncated_need = 0;      // > 0: the forward exceeded its capacity (pairs it needed): re-run before its backward
};
static bool mark_truncated(const std::weak_ptr<ViewRecord>& rec, int64_t need) {       // false: nobody is left to re-run it
  if (auto r = rec.lock()) {
    r->truncated_need = need;
    return true;
  }
  return false;
}

static bool g_provenance = true;      // renders whose inputs are new tensors of the same provenance join the open batch (SPLAT_RASTER_PROVENANCE=0: identity only)
static void set_provenance(bool on) { g_provenance = on; }

// The forwards that share one set of input tensors (the renders of a mapping iteration).  `inputs` holds the caller's tensors
// (identity decides membership), `alias` the collector's outputs the forwards consume.  A record belongs to its forward's autograd
// node (a render whose outputs are dropped frees its saved block right away, however long the batch lives); the batch only holds
// the records whose backward has been called (`armed`) until the collector has used them.
struct Batch {
  Tensor inputs[5];
  uint32_t versions[5];
  Tensor alias[5];
  std::vector<std::shared_ptr<ViewRecord>> armed;
  bool closed = false;
  int dev = 0;
  int64_t N = 0, M = 0;
  std::atomic<int> views_open{0};  // forwards of this batch whose node is alive: sizes the shared dL/dmeans2D arena
  Tensor m2d_arena;                // [views, N, 3] zeros: ONE memset for the means2D gradients of all views of the backward pass
  int64_t m2d_used = 0, m2d_slots = 0;

  // Which leaves the batch's inputs were computed from, and at which version (recorded when the batch is made): a later render
  // whose tensors are NEW objects computed the same way from the same leaves at the same versions holds the same values.
  std::vector<std::pair<const torch::autograd::Node*, uint32_t>> leaf_versions;

  // The reference's scene model builds its activations anew on every getter call (exp / sigmoid / normalize / cat:
  // /root/reference/thirdparty/gaussian_splatting/scene/gaussian_model.py:76-101) and render() calls the getters per view
  // (gaussian_renderer/__init__.py:89-111): twelve renders of one iteration hand in twelve sets of DIFFERENT tensors with IDENTICAL
  // values.  By identity they were batches of one (twelve backward passes).  They join one batch when every tensor is the batch's own
  // or has the same autograd provenance: the same chain of (pure, deterministic, whitelisted) operations over the same leaves at the
  // same versions.  The later views' own activation tensors then simply get no gradient: all of it flows through the FIRST view's
  // chain, summed over the views -- exact, because the views' chains apply the same linear map to their gradients
  // (sum_k J g_k = J sum_k g_k, up to fp32 summation order).
  // index of a whitelisted operator (-1: not on the list).  The order is that of `kPure` below: same_attributes() switches on it.
  enum PureOp { P_EXP, P_SIGMOID, P_CAT, P_REPEAT, P_DIV, P_EXPAND, P_CLAMPMIN, P_VECNORM, P_NORM1, P_TRANSPOSE, P_VIEW, P_UNSAFEVIEW,
                P_RESHAPEALIAS, P_CLONE, P_ALIAS, P_PERMUTE, P_UNSQUEEZE, P_SQUEEZE1, P_COUNT };
  static int pure_op(const std::string& n) {
    static const char* kPure[P_COUNT] = {"ExpBackward0", "SigmoidBackward0", "CatBackward0", "RepeatBackward0", "DivBackward0", "ExpandBackward0",
                                         "ClampMinBackward0", "LinalgVectorNormBackward0", "NormBackward1", "TransposeBackward0", "ViewBackward0",
                                         "UnsafeViewBackward0", "ReshapeAliasBackward0", "CloneBackward0", "AliasBackward0", "PermuteBackward0",
                                         "UnsqueezeBackward0", "SqueezeBackward1"};
    for (int i = 0; i < P_COUNT; ++i)
      if (n == kPure[i]) return i;
    return -1;
  }
  // The whitelisted operators are pure functions of their tensor inputs AND of a few non-tensor arguments (cat / norm / squeeze
  // dims, the clamp's minimum, repeat / expand / view sizes, permutations).  Two chains that differ only in such an argument -- a
  // permute of a square tensor, another eps -- have the same node names and the same topology: the arguments are compared as
  // well, and so is the shape of every intermediate result (a node's input metadata = its forward outputs).
  // (`op` identifies the node type -- both nodes carry that name -- so the casts are static: this runs ~60 times per render.)
  static bool same_syms(const std::vector<c10::SymInt>& x, const std::vector<c10::SymInt>& y) {
    if (x.size() != y.size()) return false;
    for (size_t i = 0; i < x.size(); ++i)
      if (x[i].expect_int() != y[i].expect_int()) return false;
    return true;
  }
  static bool same_scalar(const at::Scalar& x, const at::Scalar& y) {
    return x.isFloatingPoint() == y.isFloatingPoint() && x.isIntegral(true) == y.isIntegral(true) && x.toDouble() == y.toDouble();
  }
  static bool same_attributes(const torch::autograd::Node* a, const torch::autograd::Node* b, int op) {
    namespace G = torch::autograd::generated;
    if (a->num_inputs() != b->num_inputs()) return false;
    for (uint32_t i = 0; i < a->num_inputs(); ++i)
      if (a->input_metadata(i).shape_as_dim_vector() != b->input_metadata(i).shape_as_dim_vector()) return false;
#define DGR_SAME(T, EXPR)                                                                      \
  {                                                                                            \
    auto *x = static_cast<const G::T*>(a), *y = static_cast<const G::T*>(b);                   \
    return (EXPR);                                                                             \
  }
    switch (op) {
      case P_CAT: DGR_SAME(CatBackward0, x->dim == y->dim && x->tensors_size_ == y->tensors_size_)
      case P_REPEAT: DGR_SAME(RepeatBackward0, same_syms(x->repeats, y->repeats) && same_syms(x->self_sym_sizes, y->self_sym_sizes))
      case P_EXPAND: DGR_SAME(ExpandBackward0, same_syms(x->self_sym_sizes, y->self_sym_sizes))
      case P_CLAMPMIN: DGR_SAME(ClampMinBackward0, same_scalar(x->min, y->min))
      case P_VECNORM: DGR_SAME(LinalgVectorNormBackward0, x->keepdim == y->keepdim && same_scalar(x->ord, y->ord) && x->dim.list == y->dim.list)
      case P_NORM1: DGR_SAME(NormBackward1, x->keepdim == y->keepdim && x->dim == y->dim && x->p.has_value() == y->p.has_value() &&
                                                (!x->p.has_value() || same_scalar(*x->p, *y->p)))
      case P_TRANSPOSE: DGR_SAME(TransposeBackward0, x->dim0 == y->dim0 && x->dim1 == y->dim1)
      case P_VIEW: DGR_SAME(ViewBackward0, same_syms(x->self_sym_sizes, y->self_sym_sizes))
      case P_UNSAFEVIEW: DGR_SAME(UnsafeViewBackward0, same_syms(x->self_sym_sizes, y->self_sym_sizes))
      case P_RESHAPEALIAS: DGR_SAME(ReshapeAliasBackward0, same_syms(x->self_sym_sizes, y->self_sym_sizes))
      case P_PERMUTE: DGR_SAME(PermuteBackward0, x->dims == y->dims)
      case P_UNSQUEEZE: DGR_SAME(UnsqueezeBackward0, x->dim == y->dim)
      case P_SQUEEZE1: DGR_SAME(SqueezeBackward1, x->dim == y->dim && same_syms(x->self_sym_sizes, y->self_sym_sizes))
      default: return true;        // Exp, Sigmoid, Div, Clone, Alias: nothing but their tensor inputs
    }
#undef DGR_SAME
  }
  void record_leaves(const torch::autograd::Node* fn, int depth) {
    if (!fn || depth > 8) return;
    if (auto* acc = dynamic_cast<const torch::autograd::AccumulateGrad*>(fn)) {
      leaf_versions.emplace_back(fn, acc->variable._version());
      return;
    }
    for (const auto& e : fn->next_edges()) record_leaves(e.function.get(), depth + 1);
  }
  bool same_provenance(const torch::autograd::Node* a, const torch::autograd::Node* b, int depth) const {
    if (!a || !b || depth > 8) return false;
    if (a == b) {                                   // the same leaf (or a shared intermediate result)
      if (auto* acc = dynamic_cast<const torch::autograd::AccumulateGrad*>(a)) {
        for (const auto& lv : leaf_versions)
          if (lv.first == a) return acc->variable._version() == lv.second;
        return false;
      }
      return true;
    }
    const std::string na = a->name();
    const int op = pure_op(na);
    if (op < 0 || a->num_outputs() != b->num_outputs() || na != b->name()) return false;
    if (!same_attributes(a, b, op)) return false;
    for (uint32_t j = 0; j < a->num_outputs(); ++j) {
      const auto& ea = a->next_edge(j);
      const auto& eb = b->next_edge(j);
      // (an operand without a graph -- a plain tensor -- could differ in value between the two renders: not accepted)
      if (!ea.is_valid() || !eb.is_valid() || ea.input_nr != eb.input_nr) return false;
      if (!same_provenance(ea.function.get(), eb.function.get(), depth + 1)) return false;
    }
    return true;
  }
  bool matches(const Tensor* t, bool provenance) const {
    if (closed) return false;
    for (int i = 0; i < 5; ++i) {
      if (t[i].unsafeGetTensorImpl() == inputs[i].unsafeGetTensorImpl() && t[i]._version() == versions[i]) continue;
      if (!provenance || inputs[i]._version() != versions[i]) return false;
      const auto& fa = t[i].grad_fn();
      const auto& fb = inputs[i].grad_fn();
      if (!fa || !fb || t[i].output_nr() != inputs[i].output_nr() || t[i].sizes() != inputs[i].sizes() || !t[i].is_contiguous()) return false;
      if (!same_provenance(fa.get(), fb.get(), 0)) return false;
    }
    return true;
  }
  // zero-filled [N,3] for one view's dL/dmeans2D: slices of one arena per backward pass (a memset per view was 12 launches)
  Tensor take_m2d() {
    if (!m2d_arena.defined() || m2d_used >= m2d_slots) {
      m2d_slots = std::max<int64_t>(1, views_open.load());
      m2d_arena = at::zeros({m2d_slots, N, 3}, at::TensorOptions().dtype(at::kFloat).device(at::kCUDA, dev));
      m2d_used = 0;
    }
    return m2d_arena.select(0, m2d_used++);
  }
};

static bool g_prof = false;
static std::map<std::string, double> g_prof_s;
struct Tick {
  std::chrono::steady_clock::time_point t0;
  Tick() : t0(std::chrono::steady_clock::now()) {}
  void lap(const char* name) {
    if (!g_prof) return;
    const auto t1 = std::chrono::steady_clock::now();
    g_prof_s[name] += std::chrono::duration<double>(t1 - t0).count();
    t0 = t1;
  }
};

struct ForwardRun {
  std::shared_ptr<Lease> lease;
  int64_t cap = 0, R = 0;
  size_t saved_bytes = 0;
};
// One sgr_forward at the state's current capacity.  wait: synchronise on the pair count (upstream's behaviour): a forward that does
// not fit is repeated at a capacity that does -- nothing is ever dropped; otherwise asynchronous (the header comes back through the
// pinned ring, see DeviceState::post).
static ForwardRun run_forward(DeviceState& st, const SgrSettings& s, const SgrInputs& inp, const SgrOutputs& out, bool wait, hipStream_t stream) {
  const int64_t N = s.num_gaussians;
  const int H = s.image_height, W = s.image_width;
  const int64_t ntiles = (int64_t)((H + 7) / 8) * ((W + 7) / 8);
  const int64_t floor = st.floor_for(N, ntiles);
  if (st.capacity < floor) st.capacity = floor;
  ForwardRun fr;
  while (true) {
    fr.cap = st.capacity;
    const auto bytes = st.bytes_for(N, H, W, fr.cap);
    fr.saved_bytes = bytes.first;
    Tensor& scratch = st.scratch_block(0, bytes.second);
    int clean = 0;
    fr.lease = st.lease(N, H, W, fr.cap, fr.saved_bytes, &clean);
    SgrWorkspace ws = {fr.lease->block.data_ptr(), fr.saved_bytes, scratch.data_ptr(), (size_t)scratch.numel(), fr.cap, clean, (int32_t)st.longest_list};
    const int rc = sgr_forward(&s, &inp, &out, &ws, wait ? &fr.R : nullptr, (void*)stream);
    if (rc == SGR_ERR_CAPACITY) {
      st.capacity = fr.R * kCapFactor;
      fr.lease->pool.reset();                        // (layout changes with the capacity: do not hand this block back)
      continue;
    }
    check(rc, "sgr_forward");
    break;
  }
  if (wait) {
    st.last_pairs = fr.R;
    if (fr.R * kCapFactor > st.capacity) st.capacity = fr.R * kCapFactor;
  }
  ++st.forwards;
  return fr;
}

static variable_list batched_backward(Batch& batch) {
  Tick tick;
  TORCH_CHECK(!batch.closed,
              "diff_gaussian_rasterization: backward through the renders of this parameter set a second time (their workspaces were "
              "released by the first backward; render again, or set SPLAT_RASTER_BATCH=0)");
  batch.closed = true;
  for (int i = 0; i < 5; ++i)
    TORCH_CHECK(batch.inputs[i]._version() == batch.versions[i],
                "one of the variables needed for gradient computation has been modified by an inplace operation: the Gaussian "
                "parameters handed to the rasterizer changed between render and backward");
  DeviceState& st = state(batch.dev);
  std::lock_guard<std::mutex> lk(st.mu);
  if (st.batch.get() == &batch) st.batch.reset();
  std::vector<std::shared_ptr<ViewRecord>> views;
  views.swap(batch.armed);
  const int64_t N = batch.N, M = batch.M;
  c10::DeviceGuard guard(cuda_device(batch.dev));
  const hipStream_t stream = current_stream(batch.dev);
  at::AutoDispatchBelowADInplaceOrView below;
  // Every forward of this iteration has its header on the way: make sure none of them dropped pairs BEFORE producing gradients.
  // Upstream never drops a pair (it sizes its buffers inside every call, at a host synchronisation each); a forward that did
  // here -- the map's pair count jumped by more than the 4x head-room between two renders -- is RE-RUN now, synchronously, at a
  // capacity that fits: its saved block then describes the complete lists and the backward below is that of the full view.  (The
  // image the caller already consumed was rendered from truncated lists: its loss gradient is that image's.  One iteration with a
  // slightly different loss gradient for one view, and a warning, instead of an exception that ends a SLAM run: the reference loop
  // has no retry, /root/reference/src/mapper.py:426-490.)
  st.drain(true);
  int64_t rerun = 0;
  for (auto& vp : views) {
    ViewRecord& r = *vp;
    if (r.truncated_need <= 0) continue;
    const auto fopt_ = at::TensorOptions().dtype(at::kFloat).device(at::kCUDA, batch.dev);
    Tensor img = at::empty({5 * (int64_t)r.H * r.W}, fopt_);
    Tensor nt = at::empty({N}, at::TensorOptions().dtype(at::kInt).device(at::kCUDA, batch.dev));
    const int64_t hw = (int64_t)r.H * r.W;
    SgrInputs fin = {fptr(batch.alias[0]), fptr(batch.alias[2]), fptr(batch.alias[1]), nullptr, fptr(batch.alias[3]), fptr(batch.alias[4]), nullptr};
    SgrOutputs fout = {img.data_ptr<float>(), img.data_ptr<float>() + 3 * hw, img.data_ptr<float>() + 4 * hw, r.radii.data_ptr<int32_t>(),
                       nt.data_ptr<int32_t>()};
    if (st.capacity < kCapFactor * r.truncated_need) st.capacity = kCapFactor * r.truncated_need;
    ForwardRun fr = run_forward(st, r.settings, fin, fout, true, stream);
    r.lease = fr.lease;
    r.cap = fr.cap;
    r.saved_bytes = fr.saved_bytes;
    r.truncated_need = 0;
    ++rerun;
  }
  if (rerun) {
    st.reruns += rerun;
    st.unreported = std::max<int64_t>(0, st.unreported - rerun);
    TORCH_WARN("diff_gaussian_rasterization: ", rerun, " forward(s) of this backward pass had exceeded the (tile, Gaussian) pair capacity and "
               "were re-run at ", st.capacity, " pairs; the images they returned earlier were rendered from truncated tile lists");
  }
  if (st.unreported) {        // forward-only renders (nothing to re-run: their images were consumed as they were)
    TORCH_WARN("diff_gaussian_rasterization: ", st.unreported, " forward-only render(s) exceeded the pair capacity (now ", st.capacity,
               "): their images were rendered from truncated tile lists (SPLAT_RASTER_SYNC=1 sizes every forward synchronously)");
    st.unreported = 0;
  }
  tick.lap("collector: wait for the forwards' headers");
  // ONE arena for the five summed gradients (each written in full by the gather pass)
  const int64_t widths[5] = {3, 3 * M, 1, 3, 4};
  int64_t total = 0;
  for (auto w : widths) total += N * w;
  const auto fopt = at::TensorOptions().dtype(at::kFloat).device(at::kCUDA, batch.dev);
  Tensor arena = at::empty({total}, fopt);
  Tensor parts[5];
  int64_t o = 0;
  for (int i = 0; i < 5; ++i) {
    parts[i] = arena.narrow(0, o, N * widths[i]);
    o += N * widths[i];
  }
  SgrInputs inp = {fptr(batch.alias[0]), fptr(batch.alias[2]), fptr(batch.alias[1]), nullptr, fptr(batch.alias[3]), fptr(batch.alias[4]), nullptr};
  SgrGradInputs gi = {};
  gi.dL_dmeans3D = parts[0].data_ptr<float>();
  gi.dL_dshs = parts[1].data_ptr<float>();
  gi.dL_dopacities = parts[2].data_ptr<float>();
  gi.dL_dscales = parts[3].data_ptr<float>();
  gi.dL_drotations = parts[4].data_ptr<float>();
  const int nv = (int)views.size();
  Tensor dtau = at::zeros({std::max(1, nv), 6}, fopt);          // (rho[3], theta[3]) of every view
  float* dtau_ptr = dtau.data_ptr<float>();
  std::vector<SgrBackwardView> items((size_t)nv);
  std::vector<Tensor> keep;
  for (int k = 0; k < nv; ++k) {
    ViewRecord& r = *views[k];
    const auto bytes = st.bytes_for(N, r.H, r.W, r.cap);
    Tensor& sc = st.scratch_block((size_t)k, bytes.second);
    keep.push_back(sc);
    SgrBackwardView& it = items[k];
    it.settings = r.settings;
    it.radii = r.radii.data_ptr<int32_t>();
    it.ws = SgrWorkspace{r.lease->block.data_ptr(), r.saved_bytes, sc.data_ptr(), (size_t)sc.numel(), r.cap, 0, 0};
    it.dL_dcolor = r.grad_color.data_ptr<float>();
    it.dL_ddepth = fptr(r.grad_depth);
    it.dL_dmeans2D = r.m2d_ptr;
    it.dL_dtau = r.strict_pose ? r.pose_arena.data_ptr<float>() : dtau_ptr + 6 * k;
  }
  tick.lap("collector: buffers + structs");
  if (nv == 0) arena.zero_();
  else check(sgr_backward_views(nv, items.data(), &inp, &gi, (void*)stream), "sgr_backward_views");
  tick.lap("collector: sgr_backward_views (launches)");
  // The per-view gradients (means2D; pose in strict mode) were RETURNED by the views' own nodes as zero tensors before this
  // launch filled them.  Where autograd kept that very tensor as `.grad` (a fresh leaf: it steals a gradient nobody else
  // references) the values are in place now; where it copied or accumulated (an existing `.grad`, a retained non-leaf), the copy
  // holds zeros + whatever was there before: add the values (all of them in one multi-tensor launch).
  std::vector<Tensor> fix_g, fix_v;
  auto fix = [&](const Tensor& p, float* where, int64_t n) {
    if (!p.defined()) return;
    const Tensor& g = p.grad();
    if (g.defined() && g.data_ptr() != (void*)where) {
      fix_g.push_back(g);
      fix_v.push_back(at::from_blob(where, {n}, fopt).view(g.sizes()));
    }
  };
  for (int k = 0; k < nv; ++k) {
    ViewRecord& r = *views[k];
    if (r.m2d_ptr) fix(r.means2D, r.m2d_ptr, 3 * N);
    if (r.strict_pose) {
      fix(r.rho, r.pose_arena.data_ptr<float>(), 3);
      fix(r.theta, r.pose_arena.data_ptr<float>() + 3, 3);
    } else {             // deferred pose gradients: this function IS their accumulation step
      const Tensor* ps[2] = {&r.rho, &r.theta};
      for (int j = 0; j < 2; ++j) {
        const Tensor& p = *ps[j];
        if (!p.defined()) continue;
        Tensor v = dtau.select(0, k).narrow(0, 3 * j, 3).view(p.sizes());
        if (!p.grad().defined()) const_cast<Tensor&>(p).mutable_grad() = v;
        else {
          fix_g.push_back(p.grad());
          fix_v.push_back(v);
        }
      }
    }
  }
  if (!fix_g.empty()) {
    // (from_blob views keep no owner: the arenas they point into are alive until this function returns -- batch.m2d_arena, pose arenas)
    at::_foreach_add_(fix_g, fix_v);
  }
  tick.lap("collector: per-view leaf gradients");
  ++st.backwards;
  variable_list out(5);
  for (int i = 0; i < 5; ++i) out[i] = parts[i].view(batch.alias[i].sizes());
  return out;
}

// identity on (means3D, sh, opacities, scales, rotations); its backward runs once per backward pass, after every view that consumed
// its outputs has recorded its image gradients: the batched backward of all of them happens here
struct CollectNode : public torch::autograd::Node {
  std::weak_ptr<Batch> batch;        // (weak: batch -> aliases -> this node must not close a cycle that pins HBM)
  variable_list apply(variable_list&& extra) override {
    auto b = batch.lock();
    TORCH_CHECK(b, "diff_gaussian_rasterization: the renders of this backward pass are gone");
    variable_list g = batched_backward(*b);
    for (size_t i = 0; i < g.size() && i < extra.size(); ++i)       // (somebody else differentiated through the aliases: rare)
      if (extra[i].defined()) g[i] = g[i] + extra[i];
    return g;
  }
};

struct ViewNode : public torch::autograd::Node {
  std::shared_ptr<ViewRecord> rec;
  std::shared_ptr<Batch> batch;
  bool has_theta = false, has_rho = false;
  std::vector<int64_t> theta_shape, rho_shape;
  ~ViewNode() override {
    if (batch) --batch->views_open;
  }
  // (the engine calls this right after apply() unless the graph is retained: the record lives on in batch.armed until the collector
  //  has used it, and its saved block returns to the pool as soon as that is done -- not when the caller drops the loss tensor)
  void release_variables() override { rec.reset(); }
  // inputs: gradients of (color, depth, opacity); outputs follow next_edges:
  // (means3D alias, means2D, sh alias, opacities alias, scales alias, rotations alias, theta, rho)
  variable_list apply(variable_list&& g) override {
    TORCH_CHECK(rec && batch, "diff_gaussian_rasterization: backward through a render whose buffers were released");
    ViewRecord& r = *rec;
    Batch& b = *batch;
    TORCH_CHECK(!b.closed,
                "diff_gaussian_rasterization: backward through a render whose batch has already run its backward (render again, or set "
                "SPLAT_RASTER_BATCH=0 for independent per-view backward passes)");
    at::AutoDispatchBelowADInplaceOrView below;
    const auto fopt = at::TensorOptions().dtype(at::kFloat).device(at::kCUDA, b.dev);
    r.grad_color = g.size() > 0 && g[0].defined() ? f32c(g[0]) : at::zeros({3, r.H, r.W}, fopt);
    r.grad_depth = g.size() > 1 && g[1].defined() ? f32c(g[1]) : Tensor();
    // (the gradient of the `opacity` image is ignored exactly like upstream: the reference never differentiates it,
    //  slam_utils.py:71-77 / :108-119)
    variable_list out(8);
    if (r.means2D.defined()) {
      Tensor m2 = b.take_m2d();
      r.m2d_ptr = m2.data_ptr<float>();
      if (!r.m2d_deferred) out[1] = std::move(m2);
      else if (!r.means2D.grad().defined()) const_cast<Tensor&>(r.means2D).mutable_grad() = m2;   // (an existing .grad: the collector adds the values)
    } else {
      r.m2d_ptr = nullptr;
    }
    if (r.strict_pose) {
      r.pose_arena = at::zeros({6}, fopt);
      if (has_rho) out[7] = r.pose_arena.narrow(0, 0, 3).view(rho_shape);
      if (has_theta) out[6] = r.pose_arena.narrow(0, 3, 3).view(theta_shape);
    }
    if (!r.armed) {
      r.armed = true;
      b.armed.push_back(rec);
    }
    return out;
  }
};

// ---- forward ------------------------------------------------------------------------------------------------------------------------
static bool shared_ok(const Tensor& t) { return t.defined() && t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous(); }

// The reference's call (gaussian_renderer/__init__.py:130-141 at sh_degree 0: shs [N,1,3], scales + rotations, fp32 contiguous GPU
// tensors).  Returns None for anything else: the Python implementation takes those.
static py::object try_rasterize(const Tensor& means3D, const c10::optional<Tensor>& means2D_o, const c10::optional<Tensor>& sh_o,
                                const Tensor& opacities, const c10::optional<Tensor>& scales_o, const c10::optional<Tensor>& rot_o,
                                const c10::optional<Tensor>& theta_o, const c10::optional<Tensor>& rho_o, int64_t H, int64_t W, double tanfovx,
                                double tanfovy, const Tensor& bg_in, double scale_modifier, const Tensor& view_in, const Tensor& proj_in,
                                const Tensor& praw_in, int64_t sh_degree, const Tensor& campos_in, bool prefiltered, bool debug, bool sync,
                                bool defer_pose) {
  if (!sh_o || !scales_o || !rot_o || sh_degree != 0) return py::none();
  const Tensor& sh = *sh_o;
  const Tensor& scales = *scales_o;
  const Tensor& rot = *rot_o;
  if (!shared_ok(means3D) || !shared_ok(sh) || !shared_ok(opacities) || !shared_ok(scales) || !shared_ok(rot)) return py::none();
  if (sh.dim() != 3 || sh.size(1) != 1 || means3D.size(0) <= 0 || means3D.dim() != 2) return py::none();
  const int64_t N = means3D.size(0);
  if (scales.numel() != 3 * N || rot.numel() != 4 * N || opacities.numel() != N || sh.numel() != 3 * N) return py::none();
  const Tensor in5[5] = {means3D, sh, opacities, scales, rot};
  bool any_grad = false;
  for (auto& t : in5) any_grad = any_grad || t.requires_grad();
  const bool want_grad = torch::autograd::GradMode::is_enabled() && any_grad;
  // (grad mode on but only means2D / the pose deltas want a gradient: the Python path's per-view backward takes that)
  if (torch::autograd::GradMode::is_enabled() && !any_grad) {
    const bool other = (means2D_o && means2D_o->requires_grad()) || (theta_o && theta_o->requires_grad()) || (rho_o && rho_o->requires_grad());
    if (other) return py::none();
  }
  const int dev = means3D.get_device();
  c10::DeviceGuard guard(cuda_device(dev));
  const hipStream_t stream = current_stream(dev);
  DeviceState& st = state(dev);
  std::lock_guard<std::mutex> lk(st.mu);
  st.drain(false);         // (headers that have landed; the forwards of an iteration are all waited for inside their backward)
  if (st.lost) {
    // (the Python nodes raise here -- _forward / report(); a forward-only evaluation loop has no backward in which the C++ nodes
    //  could re-run the render, so it is told NOW, not in some later backward pass or never: ADVICE r4)
    const std::string msg = c10::str("diff_gaussian_rasterization: ", st.lost, " earlier render(s) without a backward pass exceeded the (tile, "
                                     "Gaussian) pair capacity (now ", st.capacity, "): the images they returned were not composited (background "
                                     "only).  Render them again; SPLAT_RASTER_SYNC=1 sizes every forward synchronously");
    st.unreported = std::max<int64_t>(0, st.unreported - st.lost);
    st.lost = 0;
    // (a Python warning -- this function is called from Python with the GIL held; TORCH_WARN outside an autograd / dispatcher frame only
    //  prints to stderr, where `warnings` filters and tests cannot see it)
    if (PyErr_WarnEx(PyExc_RuntimeWarning, msg.c_str(), 1) < 0) throw py::error_already_set();
  }

  Tensor bg, view, proj, praw, campos;
  {
    at::AutoDispatchBelowADInplaceOrView below;
    bg = f32c(bg_in); view = f32c(view_in); proj = f32c(proj_in); praw = f32c(praw_in); campos = f32c(campos_in);
  }
  TORCH_CHECK(bg.is_cuda() && view.is_cuda() && proj.is_cuda() && praw.is_cuda() && campos.is_cuda(),
              "diff_gaussian_rasterization (MI355X build): every tensor must live on the GPU; there is no CPU path");

  std::shared_ptr<Batch> b;
  if (want_grad) {
    b = st.batch;
    if (!b || !b->matches(in5, g_provenance)) {
      b = std::make_shared<Batch>();
      b->dev = dev;
      b->N = N;
      b->M = sh.size(1);
      auto cn = std::shared_ptr<CollectNode>(new CollectNode(), torch::autograd::deleteNode);
      cn->batch = b;
      cn->set_next_edges(torch::autograd::collect_next_edges(means3D, sh, opacities, scales, rot));
      for (int i = 0; i < 5; ++i) {
        b->inputs[i] = in5[i];
        b->versions[i] = in5[i]._version();
        if (in5[i].grad_fn()) b->record_leaves(in5[i].grad_fn().get(), 0);
        at::AutoDispatchBelowADInplaceOrView below;
        b->alias[i] = in5[i].detach();
      }
      for (int i = 0; i < 5; ++i) torch::autograd::create_gradient_edge(b->alias[i], cn);
      st.batch = b;
      ++st.batches;
    }
  }

  at::AutoDispatchBelowADInplaceOrView below;
  const int64_t HW = H * W;
  const auto fopt = at::TensorOptions().dtype(at::kFloat).device(at::kCUDA, dev);
  // one arena per call: colour | depth | opacity, and radii | n_touched
  Tensor fbuf = at::empty({5 * HW}, fopt);
  Tensor color = fbuf.narrow(0, 0, 3 * HW).view({3, H, W}), depth = fbuf.narrow(0, 3 * HW, HW).view({1, H, W}),
         opac = fbuf.narrow(0, 4 * HW, HW).view({1, H, W});
  Tensor ibuf = at::empty({2 * N}, at::TensorOptions().dtype(at::kInt).device(at::kCUDA, dev));
  Tensor radii = ibuf.narrow(0, 0, N), n_touched = ibuf.narrow(0, N, N);

  SgrSettings s = {(int32_t)N, (int32_t)H, (int32_t)W, 0, (int32_t)sh.size(1), (float)tanfovx, (float)tanfovy, (float)scale_modifier,
                   prefiltered ? 1 : 0, debug ? 1 : 0, fptr(bg), fptr(view), fptr(proj), fptr(praw), fptr(campos)};
  SgrInputs inp = {fptr(means3D), fptr(opacities), fptr(sh), nullptr, fptr(scales), fptr(rot), nullptr};
  SgrOutputs out = {color.data_ptr<float>(), depth.data_ptr<float>(), opac.data_ptr<float>(), radii.data_ptr<int32_t>(),
                    n_touched.data_ptr<int32_t>()};
  const std::pair<int64_t, const void*> this_map{N, means3D.data_ptr()};
  // A new map: learn its pair count before trusting the capacity.  Close to the limit (the last measured count is beyond half the
  // capacity -- never the case once a map's count has been seen, capacity = 4 x the largest): wait as well, like upstream does
  // for every forward, instead of risking a truncated image.  Otherwise asynchronous.
  // (A render that can have no backward pass -- grad mode off, evaluation -- runs asynchronously too: capacity is 4 x the largest
  //  count this map has shown, and should it be exceeded anyway the caller is warned at the next call, above.)
  const bool wait = sync || this_map != st.last_map || 2 * st.last_pairs > st.capacity;
  ForwardRun fr = run_forward(st, s, inp, out, wait, stream);
  if (wait) st.last_map = this_map;
  std::shared_ptr<Lease> lease = fr.lease;
  const int64_t cap = fr.cap;
  const size_t saved_bytes = fr.saved_bytes;
  std::shared_ptr<ViewRecord> rec;
  if (want_grad) rec = std::make_shared<ViewRecord>();
  st.post(lease->block.data_ptr(), cap, stream, rec);     // (also after a synchronous forward: the header carries the longest list)

  if (want_grad) {
    auto& r = rec;
    r->settings = s;
    r->keep[0] = bg; r->keep[1] = view; r->keep[2] = proj; r->keep[3] = praw; r->keep[4] = campos;
    r->radii = radii;
    r->lease = lease;
    r->cap = cap;
    r->saved_bytes = saved_bytes;
    r->H = (int)H;
    r->W = (int)W;
    const bool has_theta = theta_o && theta_o->defined() && theta_o->numel() == 3;
    const bool has_rho = rho_o && rho_o->defined() && rho_o->numel() == 3;
    if (means2D_o && means2D_o->defined() && means2D_o->requires_grad()) r->means2D = *means2D_o;
    if (has_theta && theta_o->requires_grad()) r->theta = *theta_o;
    if (has_rho && rho_o->requires_grad()) r->rho = *rho_o;
    // deferred leaves get no edge: their AccumulateGrad nodes are not even scheduled.  A pose delta that wants a gradient but is no
    // hook-free leaf keeps autograd's semantics for both deltas (strict mode)
    r->strict_pose = !defer_pose || (r->theta.defined() && !deferable_leaf(r->theta)) || (r->rho.defined() && !deferable_leaf(r->rho));
    r->m2d_deferred = defer_pose && deferable_leaf(r->means2D);
    auto node = std::shared_ptr<ViewNode>(new ViewNode(), torch::autograd::deleteNode);
    node->rec = r;
    node->batch = b;
    ++b->views_open;
    node->has_theta = has_theta;
    node->has_rho = has_rho;
    if (has_theta) node->theta_shape = theta_o->sizes().vec();
    if (has_rho) node->rho_shape = rho_o->sizes().vec();
    const Tensor undef;
    node->set_next_edges(torch::autograd::collect_next_edges(b->alias[0], (means2D_o && !r->m2d_deferred) ? *means2D_o : undef, b->alias[1],
                                                             b->alias[2], b->alias[3], b->alias[4],
                                                             (r->strict_pose && has_theta) ? *theta_o : undef,
                                                             (r->strict_pose && has_rho) ? *rho_o : undef));
    torch::autograd::create_gradient_edge(color, node);
    torch::autograd::create_gradient_edge(depth, node);
    torch::autograd::create_gradient_edge(opac, node);
  }
  return py::make_tuple(color, radii, depth, opac, n_touched);
}

// (saved workspace block, capacity) of the forward that produced `output` -- parity tooling (sgr_query_* read it); None if the
// output did not come from try_rasterize
static py::object saved_block_of(const Tensor& output) {
  auto fn = output.grad_fn();
  auto* vn = dynamic_cast<ViewNode*>(fn.get());
  if (!vn || !vn->rec) return py::none();
  return py::make_tuple(vn->rec->lease->block, vn->rec->cap);
}

// would a render that hands in `t` join a batch that was opened with `ref` in the same argument slot?  provenance_open(ref) plays
// the first render of an iteration (the leaves' versions are recorded THEN), provenance_joins(t) a later one.  CPU-testable: no kernel.
static std::unique_ptr<Batch> g_probe;
static void provenance_open(const Tensor& ref) {
  g_probe = std::make_unique<Batch>();
  for (int i = 0; i < 5; ++i) {
    g_probe->inputs[i] = ref;
    g_probe->versions[i] = ref._version();
  }
  if (ref.grad_fn()) g_probe->record_leaves(ref.grad_fn().get(), 0);
}
static bool provenance_joins(const Tensor& t) {
  TORCH_CHECK(g_probe, "provenance_open first");
  const Tensor in5[5] = {t, t, t, t, t};
  return g_probe->matches(in5, true);
}

static void check_overflow() {
  std::vector<DeviceState*> sts;
  {
    std::lock_guard<std::mutex> lk(g_states_mu);
    for (auto& kv : g_states) sts.push_back(kv.second.get());
  }
  for (auto* st : sts) {
    std::lock_guard<std::mutex> lk(st->mu);
    st->report(true);
  }
}

static py::dict stats(int64_t dev) {
  DeviceState& st = state((int)dev);
  std::lock_guard<std::mutex> lk(st.mu);
  py::dict d;
  d["capacity"] = st.capacity;
  d["last_pairs"] = st.last_pairs;
  d["longest_list"] = st.longest_list;
  d["overflowed"] = st.overflowed;
  d["forwards"] = st.forwards;
  d["batches"] = st.batches;
  d["backwards"] = st.backwards;
  d["pending"] = (int64_t)st.pending.size();
  d["reruns"] = st.reruns;
  d["unreported"] = st.unreported;
  int64_t blocks = 0;
  for (auto& kv : st.pools) blocks += (int64_t)kv.second->blocks.size();
  d["pool_blocks"] = blocks;
  return d;
}
// tests: shrink the capacity below what the map needs (the overflow protocol must then recover)
static void set_capacity(int64_t dev, int64_t cap, bool forget_map, int64_t floor_override, int64_t last_pairs) {
  DeviceState& st = state((int)dev);
  std::lock_guard<std::mutex> lk(st.mu);
  st.capacity = cap;
  st.floor_override = floor_override;
  if (last_pairs >= 0) st.last_pairs = last_pairs;
  if (forget_map) st.last_map = {-1, nullptr};
  st.pools.clear();
}

// ---- fused mapping loss (slam_utils.py:71-105; SSIM branch off): one pass, loss + the four gradients ------------------------------
struct LossNode : public torch::autograd::Node {
  Tensor arena;                      // dL/dimage | dL/ddepth | d/da, d/db | loss | scratch
  int64_t H = 0, W = 0;
  std::vector<int64_t> dshape;
  bool has_exp = false;
  Tensor defer_a, defer_b;            // exposure leaves whose `.grad` this node writes itself (deferable_leaf)
  void release_variables() override { arena.reset(); }
  variable_list apply(variable_list&& g) override {
    TORCH_CHECK(arena.defined(), "mapping loss: backward through the graph a second time");
    at::AutoDispatchBelowADInplaceOrView below;
    const int64_t hw = H * W;
    Tensor s = arena.narrow(0, 0, 4 * hw + 2) * g[0];          // the four gradients are contiguous: ONE scaling launch
    variable_list out(4);
    out[0] = s.narrow(0, 0, 3 * hw).view({3, H, W});
    out[1] = s.narrow(0, 3 * hw, hw).view(dshape);
    if (has_exp) {
      if (defer_a.defined()) deposit_grad(defer_a, s.narrow(0, 4 * hw, 1).view(defer_a.sizes()));
      else out[2] = s.narrow(0, 4 * hw, 1);
      if (defer_b.defined()) deposit_grad(defer_b, s.narrow(0, 4 * hw + 1, 1).view(defer_b.sizes()));
      else out[3] = s.narrow(0, 4 * hw + 1, 1);
    }
    return out;
  }
};

static Tensor mapping_loss(const Tensor& image_in, const Tensor& depth_in, const c10::optional<Tensor>& exp_a, const c10::optional<Tensor>& exp_b,
                           const Tensor& gt_image, const Tensor& gt_depth, double alpha, double thr, bool defer_leaf_grads) {
  TORCH_CHECK(image_in.is_cuda() && image_in.dim() == 3, "mapping_loss: image must be a [3,H,W] GPU tensor");
  const int dev = image_in.get_device();
  c10::DeviceGuard guard(cuda_device(dev));
  const hipStream_t stream = current_stream(dev);
  const int64_t H = image_in.size(1), W = image_in.size(2), hw = H * W;
  const bool has_exp = exp_a && exp_a->defined();
  Tensor arena, loss, image, depth, gti, gtd;
  {
    at::AutoDispatchBelowADInplaceOrView below;
    image = f32c(image_in);
    depth = f32c(depth_in);
    gti = f32c(gt_image);
    gtd = f32c(gt_depth);
    arena = at::empty({4 * hw + 4 + 4096}, at::TensorOptions().dtype(at::kFloat).device(at::kCUDA, dev));
    float* a = arena.data_ptr<float>();
    check(sgr_mapping_loss((int32_t)H, (int32_t)W, image.data_ptr<float>(), depth.data_ptr<float>(), gti.data_ptr<float>(), gtd.data_ptr<float>(),
                           has_exp ? exp_a->data_ptr<float>() : nullptr, has_exp ? exp_b->data_ptr<float>() : nullptr, (float)alpha, (float)thr,
                           1.0f, a + 4 * hw + 2, a, a + 3 * hw, a + 4 * hw, a + 4 * hw + 1, a + 4 * hw + 4, 4 * 4096, (void*)stream),
          "sgr_mapping_loss");
    loss = arena.select(0, 4 * hw + 2);          // 0-dim (`view({})` would pick the view(ScalarType) overload)
  }
  const bool need = torch::autograd::GradMode::is_enabled() &&
                    (image_in.requires_grad() || depth_in.requires_grad() || (has_exp && (exp_a->requires_grad() || exp_b->requires_grad())));
  if (need) {
    auto node = std::shared_ptr<LossNode>(new LossNode(), torch::autograd::deleteNode);
    node->arena = arena;
    node->H = H;
    node->W = W;
    node->dshape = depth_in.sizes().vec();
    node->has_exp = has_exp;
    const Tensor undef;
    if (has_exp && defer_leaf_grads) {
      if (deferable_leaf(*exp_a) && exp_a->numel() == 1) node->defer_a = *exp_a;
      if (exp_b && deferable_leaf(*exp_b) && exp_b->numel() == 1) node->defer_b = *exp_b;
    }
    node->set_next_edges(torch::autograd::collect_next_edges(image_in, depth_in, (has_exp && !node->defer_a.defined()) ? *exp_a : undef,
                                                             (has_exp && !node->defer_b.defined()) ? *exp_b : undef));
    torch::autograd::create_gradient_edge(loss, node);
  }
  return loss;
}

// ---- optimiser step of one parameter group (FusedAdam.step): same arithmetic as sgr_adam_step from Python, without a ctypes call,
// a `.item()` and an in-place CPU add per tensor --------------------------------------------------------------------------------------
static void adam_group_step(const std::vector<Tensor>& params, const std::vector<Tensor>& grads, const std::vector<Tensor>& exp_avgs,
                            const std::vector<Tensor>& exp_avg_sqs, const std::vector<Tensor>& steps, double lr, double b1, double b2,
                            double eps) {
  const size_t n = params.size();
  TORCH_CHECK(grads.size() == n && exp_avgs.size() == n && exp_avg_sqs.size() == n && steps.size() == n, "adam_group_step: ragged lists");
  if (n == 0) return;
  const int dev = params[0].get_device();
  c10::DeviceGuard guard(cuda_device(dev));
  const hipStream_t stream = current_stream(dev);
  std::vector<SgrAdamTensor> small;
  std::vector<Tensor> keep;
  for (size_t i = 0; i < n; ++i) {
    const Tensor& p = params[i];
    TORCH_CHECK(steps[i].device().is_cpu() && steps[i].scalar_type() == at::kFloat && steps[i].numel() == 1,
                "adam_group_step: `step` must be a one-element fp32 CPU tensor (what torch.optim.Adam creates)");
    float* sp = steps[i].data_ptr<float>();
    sp[0] += 1.0f;                                  // state["step"] += 1
    const int64_t step = (int64_t)sp[0];
    const int64_t cnt = p.numel();
    if (cnt == 0) continue;
    Tensor g = grads[i].is_contiguous() ? grads[i] : grads[i].contiguous();
    // the kernels write through raw pointers: bump the version counter like torch.optim.Adam's in-place ops do
    p.unsafeGetTensorImpl()->bump_version();
    if (cnt <= 4096 && n > 1) {
      small.push_back(SgrAdamTensor{p.data_ptr<float>(), g.data_ptr<float>(), exp_avgs[i].data_ptr<float>(), exp_avg_sqs[i].data_ptr<float>(), cnt, step});
      keep.push_back(g);
      continue;
    }
    check(sgr_adam_step(cnt, p.data_ptr<float>(), g.data_ptr<float>(), exp_avgs[i].data_ptr<float>(), exp_avg_sqs[i].data_ptr<float>(), (float)lr,
                        (float)b1, (float)b2, (float)eps, step, (void*)stream),
          "sgr_adam_step");
  }
  if (!small.empty())
    check(sgr_adam_step_multi((int32_t)small.size(), small.data(), (float)lr, (float)b1, (float)b2, (float)eps, (void*)stream), "sgr_adam_step_multi");
}

using AdamGroupArgs = std::tuple<std::vector<Tensor>, std::vector<Tensor>, std::vector<Tensor>, std::vector<Tensor>, std::vector<Tensor>, double, double,
                                 double, double>;
// every parameter group of an optimiser in one call
static void adam_groups_step(const std::vector<AdamGroupArgs>& groups) {
  for (const auto& g : groups)
    adam_group_step(std::get<0>(g), std::get<1>(g), std::get<2>(g), std::get<3>(g), std::get<4>(g), std::get<5>(g), std::get<6>(g), std::get<7>(g),
                    std::get<8>(g));
}

// add_densification_stats + the max_radii2D update of SEVERAL views (gaussian_model.py:738-742, src/mapper.py:522-529): one host call
static void densify_stats_views(const std::vector<Tensor>& means2D_grads, const std::vector<Tensor>& radii, Tensor accum, Tensor denom,
                                Tensor max_radii) {
  TORCH_CHECK(means2D_grads.size() == radii.size(), "densify_stats_views: ragged lists");
  if (radii.empty()) return;
  const int dev = accum.get_device();
  c10::DeviceGuard guard(cuda_device(dev));
  const hipStream_t stream = current_stream(dev);
  const int64_t n = radii[0].numel();
  for (size_t i = 0; i < radii.size(); ++i) {
    TORCH_CHECK(means2D_grads[i].is_contiguous() && means2D_grads[i].scalar_type() == at::kFloat && means2D_grads[i].numel() == 3 * n &&
                    radii[i].scalar_type() == at::kInt && radii[i].numel() == n,
                "densify_stats_views: [N,3] fp32 gradients and [N] int32 radii expected");
    check(sgr_densify_stats(n, means2D_grads[i].data_ptr<float>(), radii[i].data_ptr<int32_t>(), accum.data_ptr<float>(), denom.data_ptr<float>(),
                            max_radii.data_ptr<float>(), (void*)stream),
          "sgr_densify_stats");
  }
}

static void profile_enable(bool on) {
  g_prof = on;
  g_prof_s.clear();
}
static py::dict profile_read() {
  py::dict d;
  for (auto& kv : g_prof_s) d[py::str(kv.first)] = kv.second;
  g_prof_s.clear();
  return d;
}

}  // namespace dgr

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "C++ autograd nodes + workspace state of the MI355X drop-in diff_gaussian_rasterization (host only; kernels: libsplat_hip.so)";
  m.def("try_rasterize", &dgr::try_rasterize, py::arg("means3D"), py::arg("means2D"), py::arg("shs"), py::arg("opacities"), py::arg("scales"),
        py::arg("rotations"), py::arg("theta"), py::arg("rho"), py::arg("image_height"), py::arg("image_width"), py::arg("tanfovx"),
        py::arg("tanfovy"), py::arg("bg"), py::arg("scale_modifier"), py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("projmatrix_raw"),
        py::arg("sh_degree"), py::arg("campos"), py::arg("prefiltered"), py::arg("debug"), py::arg("sync"), py::arg("defer_pose"));
  m.def("saved_block_of", &dgr::saved_block_of);
  m.def("check_overflow", &dgr::check_overflow);
  m.def("set_provenance", &dgr::set_provenance);
  m.def("provenance_open", &dgr::provenance_open);
  m.def("provenance_joins", &dgr::provenance_joins);
  m.def("stats", &dgr::stats);
  m.def("set_capacity", &dgr::set_capacity, py::arg("device"), py::arg("capacity"), py::arg("forget_map") = false, py::arg("floor_override") = -1,
        py::arg("last_pairs") = -1);
  m.def("mapping_loss", &dgr::mapping_loss);
  m.def("adam_group_step", &dgr::adam_group_step);
  m.def("adam_groups_step", &dgr::adam_groups_step);
  m.def("densify_stats_views", &dgr::densify_stats_views);
  m.def("profile_enable", &dgr::profile_enable);
  m.def("profile_read", &dgr::profile_read);
  m.def("abi_version", []() { return sgr_abi_version(); });
}
