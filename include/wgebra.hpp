// wgebra.hpp -- header-only C++17 mirror of the wgcore / wgebra surface the dense path uses, over the C ABI
// (wgebra_hip.h).  Same type and method names, parameter order and error behaviour as the reference:
//   wgcore::GpuInstance      crates/wgcore/src/gpu.rs:7-78
//   wgcore::TensorBuilder / GpuTensor<T> / GpuTensorView<T>   crates/wgcore/src/tensor.rs:41-706
//   wgcore::ViewShape / ViewShapeBuffers                      crates/wgcore/src/shapes.rs:9-117
//   wgcore::CommandEncoder / ComputePass / Queue              crates/wgcore/src/kernel.rs:7-27 (+ wgpu)
//   wgebra::Gemm / Gemv / Reduce / OpAssign (+ variant enums) crates/wgebra/src/linalg/{gemm,gemv,reduce,op_assign}.rs
// Where the reference panics (assert_eq!), these throw wgcore::Panic (a std::logic_error) carrying the same message.
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "wgebra_hip.h"

namespace wgcore {

struct Panic : std::logic_error { int status; Panic(int s, const std::string &m) : std::logic_error(m), status(s) {} };
struct Error : std::runtime_error { int status; Error(int s, const std::string &m) : std::runtime_error(m), status(s) {} };

inline void check(int rc) {
    if (rc == WG_OK) return;
    std::string msg = wg_last_error_string();
    if (rc == WG_ERR_DIM_MISMATCH || rc == WG_ERR_PRECONDITION) throw Panic(rc, msg);
    throw Error(rc, msg);
}

// wgpu::BufferUsages
struct BufferUsages {
    enum : uint32_t { MAP_READ = 1, MAP_WRITE = 2, COPY_SRC = 4, COPY_DST = 8, INDEX = 16, VERTEX = 32, UNIFORM = 64,
                      STORAGE = 128, INDIRECT = 256, QUERY_RESOLVE = 512 };
};

using ViewShape = wg_view_shape; // byte-identical (shapes.rs:9-21)

// shapes.rs:45-117: a cache of uniform buffers in the reference; shapes are kernel arguments here, so it is empty and
// exists for signature compatibility.
struct ViewShapeBuffers { static ViewShapeBuffers create() { return {}; } };

template <typename T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr wg_dtype value = WG_F32; };
#if defined(__FLT16_MANT_DIG__)
template <> struct dtype_of<_Float16> { static constexpr wg_dtype value = WG_F16; };
#endif

// Everything that owns a handle created on a context (tensors, command buffers) shares ownership of that context: wg_buf_destroy /
// wg_cmdbuf_destroy dereference it, so it must be destroyed last -- like wgpu buffers keep their wgpu::Device alive.
using CtxPtr = std::shared_ptr<wg_ctx>;

class Device {
  public:
    explicit Device(CtxPtr c) : ctx_(std::move(c)) {}
    wg_ctx *raw() const { return ctx_.get(); }
    const CtxPtr &shared() const { return ctx_; }
  private:
    CtxPtr ctx_;
};

class CommandBuffer {
  public:
    CommandBuffer() = default;
    CommandBuffer(wg_cmdbuf *cb, CtxPtr ctx) : cb_(cb, [ctx](wg_cmdbuf *p) { wg_cmdbuf_destroy(p); }) {}
    wg_cmdbuf *raw() const { return cb_.get(); }
  private:
    std::shared_ptr<wg_cmdbuf> cb_;
};

// timestamps.rs:9-248: a pool of timestamp slots, written at compute-pass boundaries or explicitly inside a pass; events on the context's stream
class ComputePass;
class CommandEncoder;
class GpuTimestamps {
  public:
    struct PassWrites { GpuTimestamps *query_set; uint32_t beginning_of_pass_write_index, end_of_pass_write_index; }; // wgpu::ComputePassTimestampWrites
    GpuTimestamps(wg_ctx *ctx, CtxPtr keep, uint32_t capacity) : keep_(std::move(keep)), capacity_(capacity) {
        wg_timestamps *t = nullptr;
        check(wg_timestamps_create(ctx, capacity, &t));
        ts_.reset(t, [](wg_timestamps *p) { wg_timestamps_destroy(p); });
    }
    bool is_empty() const { return len() == 0; }
    uint32_t len() const { return wg_timestamps_len(ts_.get()); }
    wg_timestamps *query_set() const { return ts_.get(); }
    void clear() { check(wg_timestamps_clear(ts_.get())); }
    // all or none: std::nullopt-like `false` when the slots do not fit (timestamps.rs:59-94)
    bool next_query_indices(uint32_t count, uint32_t *first) { check(wg_timestamps_reserve(ts_.get(), count, first)); return *first != UINT32_MAX; }
    bool next_query_index(uint32_t *index) { return next_query_indices(1, index); }
    bool next_compute_pass_timestamp_writes(PassWrites *w) {
        uint32_t first = 0;
        if (!next_query_indices(2, &first)) return false;
        *w = PassWrites{ this, first, first + 1 };
        return true;
    }
    inline bool write_next_timestamp(ComputePass &pass, uint32_t *index = nullptr);
    inline bool write_timestamp_at(ComputePass &pass, uint32_t query_index);
    void resolve(CommandEncoder &) const {} // events need no resolve copy
    std::vector<double> wait_for_results_ms() const {
        std::vector<double> ms(len());
        check(wg_timestamps_wait_for_results_ms(ts_.get(), ms.data(), (uint32_t)ms.size()));
        return ms;
    }
    std::vector<uint64_t> wait_for_results() const { // raw values: nanoseconds since the first written slot (period 1)
        std::vector<uint64_t> raw;
        for (double m : wait_for_results_ms()) raw.push_back((uint64_t)(m * 1.0e6 + 0.5));
        return raw;
    }
    static std::vector<double> timestamps_to_ms(const std::vector<uint64_t> &raw, float timestamp_period) {
        std::vector<double> ms;
        for (uint64_t t : raw) ms.push_back((double)t * (double)timestamp_period / 1.0e6);
        return ms;
    }
  private:
    CtxPtr keep_;
    std::shared_ptr<wg_timestamps> ts_;
    uint32_t capacity_;
};

class CommandEncoder;
class ComputePass {
  public:
    explicit ComputePass(wg_ctx *c) : ctx_(c) {}
    ComputePass(wg_ctx *c, GpuTimestamps::PassWrites w) : ctx_(c), writes_(w), timed_(true) { // beginning_of_pass_write_index
        check(wg_timestamps_write_at(ctx_, writes_.query_set->query_set(), writes_.beginning_of_pass_write_index));
    }
    ComputePass(ComputePass &&o) noexcept : ctx_(o.ctx_), writes_(o.writes_), timed_(o.timed_) { o.timed_ = false; }
    ComputePass(const ComputePass &) = delete;
    ~ComputePass() { if (timed_) (void)wg_timestamps_write_at(ctx_, writes_.query_set->query_set(), writes_.end_of_pass_write_index); } // drop(pass): end_of_pass_write_index
    wg_ctx *ctx() const { return ctx_; }
  private:
    wg_ctx *ctx_;
    GpuTimestamps::PassWrites writes_{};
    bool timed_ = false;
};
inline bool GpuTimestamps::write_next_timestamp(ComputePass &pass, uint32_t *index) {
    uint32_t i = 0;
    if (!next_query_index(&i)) return false;
    check(wg_timestamps_write_at(pass.ctx(), ts_.get(), i));
    if (index) *index = i;
    return true;
}
inline bool GpuTimestamps::write_timestamp_at(ComputePass &pass, uint32_t query_index) {
    if (query_index >= capacity_) return false;
    check(wg_timestamps_write_at(pass.ctx(), ts_.get(), query_index));
    return true;
}

// wgpu::CommandEncoder + CommandEncoderExt::compute_pass (kernel.rs:15-27). record == false: work is enqueued as encoded.
class CommandEncoder {
  public:
    CommandEncoder(CtxPtr c, bool record) : ctx_(std::move(c)), record_(record) { if (record_) check(wg_encoder_begin(ctx_.get())); }
    ComputePass compute_pass(const char * /*label*/, std::nullptr_t = nullptr) { return ComputePass(ctx_.get()); }
    ComputePass compute_pass(const char * /*label*/, GpuTimestamps &timestamps) { // kernel.rs:15-27: the pass writes the next two slots at its beginning and its end
        GpuTimestamps::PassWrites w{};
        if (!timestamps.next_compute_pass_timestamp_writes(&w)) return ComputePass(ctx_.get());
        return ComputePass(ctx_.get(), w);
    }
    CommandBuffer finish() {
        if (!record_) return CommandBuffer();
        wg_cmdbuf *cb = nullptr;
        check(wg_encoder_finish(ctx_.get(), &cb));
        return CommandBuffer(cb, ctx_);
    }
    wg_ctx *ctx() const { return ctx_.get(); }
  private:
    CtxPtr ctx_;
    bool record_;
};

class Queue {
  public:
    explicit Queue(wg_ctx *c) : ctx_(c) {}
    void submit(const CommandBuffer &cb) const { if (cb.raw()) check(wg_queue_submit(ctx_, cb.raw())); }
    float get_timestamp_period() const { return 1.f; } // nanoseconds per tick of GpuTimestamps::wait_for_results' raw values
  private:
    wg_ctx *ctx_;
};

// gpu.rs:7-78
class GpuInstance {
  public:
    static GpuInstance create(int device_index = 0) { // GpuInstance::new().await
        wg_ctx *c = nullptr;
        check(wg_ctx_create(device_index, &c));
        return GpuInstance(c);
    }
    static GpuInstance with_backends(unsigned /*backends*/, int device_index = 0) { return create(device_index); } // gpu.rs: one backend here
    static GpuInstance without_gl(int device_index = 0) { return create(device_index); }
    const Device &device() const { return device_; }
    const Device &device_arc() const { return device_; } // gpu.rs: the device behind an Arc -- the context is shared already
    const Queue &queue() const { return queue_; }
    GpuTimestamps timestamps(uint32_t capacity) const { return GpuTimestamps(ctx_.get(), ctx_, capacity); } // GpuTimestamps::new(device, capacity)
    CommandEncoder create_command_encoder(bool record = false) const { return CommandEncoder(ctx_, record); }
    void poll_wait() const { check(wg_ctx_sync(ctx_.get())); }
  private:
    explicit GpuInstance(wg_ctx *c) : ctx_(c, [](wg_ctx *p) { wg_ctx_destroy(p); }), device_(ctx_), queue_(c) {}
    CtxPtr ctx_; // declared first: device_ copies it
    Device device_;
    Queue queue_;
};

template <typename T> class GpuTensor;

// tensor.rs:415-542: (ViewShape, &Buffer), Copy
template <typename T>
class GpuTensorView {
  public:
    GpuTensorView(ViewShape s, wg_buf *b) : shape_(s), buf_(b) {}
    ViewShape shape() const { return shape_; }
    wg_buf *buffer() const { return buf_; }
    uint32_t len() const { return shape_.size[0]; }
    GpuTensorView rows(uint32_t first, uint32_t nrows) const { // tensor.rs:445-462 / 498-510
        ViewShape s = shape_;
        s.size[0] = nrows;
        s.offset += first;
        return GpuTensorView(s, buf_);
    }
    GpuTensorView columns(uint32_t first_col, uint32_t ncols) const { // tensor.rs:484-496
        ViewShape s = shape_;
        s.size[1] = ncols;
        s.size[2] = 1;
        s.offset += s.stride * first_col;
        return GpuTensorView(s, buf_);
    }
    GpuTensorView matrix(uint32_t id) const { // tensor.rs:466-480 (stride_mat: 1, as in the reference)
        if (id >= shape_.size[2]) throw Panic(WG_ERR_PRECONDITION, "assertion failed: matrix_id < nmats");
        ViewShape s = shape_;
        s.size[2] = 1;
        s.offset += s.stride_mat * id;
        s.stride_mat = 1;
        return GpuTensorView(s, buf_);
    }
  private:
    ViewShape shape_;
    wg_buf *buf_;
};

// tensor.rs:192-400; DIM is a run-time property here (0..3)
template <typename T>
class GpuTensor {
  public:
    GpuTensor(CtxPtr ctx, wg_buf *b, std::vector<uint32_t> shape)
        : ctx_(std::move(ctx)), buf_(b, [](wg_buf *p) { wg_buf_destroy(p); }), shape_(std::move(shape)) {}
    uint64_t len() const { uint64_t n = 1; for (auto s : shape_) n *= s; return n; }
    uint64_t bytes_len() const { return sizeof(T) * len(); }
    const std::vector<uint32_t> &shape() const { return shape_; }
    wg_buf *buffer() const { return buf_.get(); }

    void copy_from(CommandEncoder &enc, const GpuTensor &src) const { // tensor.rs:227-233
        if (len() != src.len()) throw Panic(WG_ERR_DIM_MISMATCH, "assertion `left == right` failed (copy_from: len)");
        check(wg_buf_copy(enc.ctx(), src.buffer(), 0, buffer(), 0, bytes_len()));
    }
    std::vector<T> read(const Device &dev) const { // tensor.rs:375-384 (blocking)
        std::vector<T> out(len());
        check(wg_buf_read(dev.raw(), buffer(), 0, out.data(), bytes_len()));
        return out;
    }
    // as_embedded_view::<_, 3>() (tensor.rs:287-297): column-major defaults
    GpuTensorView<T> as_embedded_view() const {
        ViewShape s{{1, 1, 1}, 1, 1, 0};
        for (size_t i = 0; i < shape_.size() && i < 3; ++i) s.size[i] = shape_[i];
        s.stride = shape_.empty() ? 1 : shape_[0];
        s.stride_mat = (shape_.empty() ? 1 : shape_[0]) * (shape_.size() > 1 ? shape_[1] : 1);
        return GpuTensorView<T>(s, buffer());
    }
    operator GpuTensorView<T>() const { return as_embedded_view(); } // From<&GpuTensor> (tensor.rs:403-409)
    GpuTensorView<T> as_view() const { return as_embedded_view(); }   // tensor.rs:282-284
    bool is_empty() const { return len() == 0; }                       // tensor.rs:198-200
    uint64_t bytes_len_encased() const { return bytes_len(); }         // tensor.rs:217-222: T::min_size() * len; the element types here are scalars (size == min_size)
    void copy_from_encased(CommandEncoder &enc, const GpuTensor &src) const { copy_from(enc, src); } // tensor.rs:235-241
    void copy_from_view(CommandEncoder &enc, GpuTensorView<T> src) const {                           // tensor.rs:244-264: `bytes_len()` bytes from the view's offset on
        const uint32_t rows = shape_.empty() ? 1u : shape_[0];
        if (src.shape().size[0] != rows) throw Panic(WG_ERR_DIM_MISMATCH, "assertion `left == right` failed (copy_from_view: rows)");
        check(wg_buf_copy(enc.ctx(), src.buffer(), (uint64_t)src.shape().offset * sizeof(T), buffer(), 0, bytes_len()));
    }
    // tensor.rs:514-541: a view of the first prod(shape) elements with the given (default: column-major) strides
    GpuTensorView<T> reshape(const std::vector<uint32_t> &shape, int64_t stride = -1, int64_t stride_mat = -1) const {
        uint32_t n = 1, m = 1;
        for (auto v : shape) n *= v; // (the reference multiplies in u32)
        for (auto v : shape_) m *= v;
        if (n > m) throw Panic(WG_ERR_PRECONDITION, "assertion failed: reshape to more elements than the tensor holds");
        ViewShape s{{1, 1, 1}, 1, 1, 0};
        for (size_t i = 0; i < shape.size() && i < 3; ++i) s.size[i] = shape[i];
        const uint32_t s0 = shape.size() > 0 ? shape[0] : 1u, s1 = shape.size() > 1 ? shape[1] : 1u;
        s.stride = stride < 0 ? s0 : (uint32_t)stride;
        s.stride_mat = stride_mat < 0 ? s0 * s1 : (uint32_t)stride_mat;
        return GpuTensorView<T>(s, buffer());
    }
    // GpuMatrix::slice (tensor.rs:587-594): offset = i + j * nrows with the SLICE's nrows, as the reference computes it (columns / rows give the conventional blocks)
    GpuTensorView<T> slice(uint32_t i, uint32_t j, uint32_t nrows, uint32_t ncols) const {
        if (shape_.size() != 2) throw Panic(WG_ERR_PRECONDITION, "GpuMatrix::slice on a tensor that is not a matrix");
        return GpuTensorView<T>(ViewShape{{nrows, ncols, 1}, shape_[0], shape_[0] * shape_[1], i + j * nrows}, buffer());
    }
    // tensor.rs:277-279: gives up the tensor, keeps the allocation (shared here: the buffer lives as long as the last holder)
    std::shared_ptr<wg_buf> into_inner() && { return std::move(buf_); }
    // GpuMatrix / GpuVector sugar (tensor.rs:544-706)
    static GpuTensor uninit(const Device &dev, std::vector<uint32_t> shape, uint32_t usage);
    static GpuTensor init(const Device &dev, std::vector<uint32_t> shape, const std::vector<T> &data, uint32_t usage);
    static GpuTensor uninit_encased(const Device &dev, std::vector<uint32_t> shape, uint32_t usage) { return uninit(dev, std::move(shape), usage); }
    static GpuTensor encase(const Device &dev, const std::vector<T> &data, uint32_t usage) { return init(dev, { (uint32_t)data.size() }, data, usage); }
  private:
    CtxPtr ctx_;                  // declared before buf_: members are destroyed in reverse order, so the buffer goes first
    std::shared_ptr<wg_buf> buf_;
    std::vector<uint32_t> shape_;
};
template <typename T> using GpuScalar = GpuTensor<T>;
template <typename T> using GpuVector = GpuTensor<T>;
template <typename T> using GpuMatrix = GpuTensor<T>;
template <typename T> using GpuCube = GpuTensor<T>;

// tensor.rs:65-187
class TensorBuilder {
  public:
    static TensorBuilder scalar(uint32_t usage) { return TensorBuilder({}, usage); }
    static TensorBuilder vector(uint32_t dim, uint32_t usage) { return TensorBuilder({dim}, usage); }
    static TensorBuilder matrix(uint32_t nrows, uint32_t ncols, uint32_t usage) { return TensorBuilder({nrows, ncols}, usage); }
    static TensorBuilder tensor(std::vector<uint32_t> shape, uint32_t usage) { return TensorBuilder(std::move(shape), usage); }
    uint64_t len() const { uint64_t n = 1; for (auto s : shape_) n *= s; return n; }

    template <typename T> GpuTensor<T> build(const Device &dev) const {
        wg_buf *b = nullptr;
        check(wg_buf_create(dev.raw(), sizeof(T) * len(), usage_, &b));
        return GpuTensor<T>(dev.shared(), b, shape_);
    }
    template <typename T> GpuTensor<T> build_init(const Device &dev, const std::vector<T> &data) const {
        if (data.size() < len()) // tensor.rs:176-182
            throw Panic(WG_ERR_PRECONDITION, "Incorrect number of elements provided for initializing Tensor.Expected at least " +
                                                 std::to_string(len()) + ", found " + std::to_string(data.size()));
        wg_buf *b = nullptr;
        check(wg_buf_create_init(dev.raw(), data.data(), sizeof(T) * len(), usage_, &b));
        return GpuTensor<T>(dev.shared(), b, shape_);
    }
    // tensor.rs:148-173: raw bytes; items in their storage layout (the scalar element types of this path: the same bytes as build_init); uninitialised, sized by the layout
    template <typename T> GpuTensor<T> build_bytes(const Device &dev, const void *data, size_t nbytes) const {
        wg_buf *b = nullptr;
        check(wg_buf_create_init(dev.raw(), data, nbytes, usage_, &b));
        return GpuTensor<T>(dev.shared(), b, shape_);
    }
    template <typename T> GpuTensor<T> build_encase(const Device &dev, const std::vector<T> &data) const { return build_init<T>(dev, data); }
    template <typename T> GpuTensor<T> build_uninit_encased(const Device &dev) const { return build<T>(dev); }
  private:
    TensorBuilder(std::vector<uint32_t> shape, uint32_t usage) : shape_(std::move(shape)), usage_(usage) {}
    std::vector<uint32_t> shape_;
    uint32_t usage_;
};

template <typename T> GpuTensor<T> GpuTensor<T>::uninit(const Device &dev, std::vector<uint32_t> shape, uint32_t usage) {
    return TensorBuilder::tensor(std::move(shape), usage).template build<T>(dev);
}
template <typename T> GpuTensor<T> GpuTensor<T>::init(const Device &dev, std::vector<uint32_t> shape, const std::vector<T> &data, uint32_t usage) {
    return TensorBuilder::tensor(std::move(shape), usage).template build_init<T>(dev, data);
}

} // namespace wgcore

namespace wgebra {
using namespace wgcore;

enum class GemmVariant { Gemm = WG_GEMM, GemmFast = WG_GEMM_FAST, GemmTr = WG_GEMM_TR, GemmTrFast = WG_GEMM_TR_FAST };
enum class GemvVariant { Gemv = WG_GEMV, GemvFast = WG_GEMV_FAST, GemvTr = WG_GEMV_TR, GemvTrFast = WG_GEMV_TR_FAST };
enum class ReduceOp { Min = WG_REDUCE_MIN, Max = WG_REDUCE_MAX, Sum = WG_REDUCE_SUM, Prod = WG_REDUCE_PROD, SqNorm = WG_REDUCE_SQNORM };
enum class OpAssignVariant { Add = WG_OP_ADD, Sub = WG_OP_SUB, Mul = WG_OP_MUL, Div = WG_OP_DIV, Copy = WG_OP_COPY };

// linalg/shape.rs:11-15: the shader definitions that make `Shape` row-major; pass them to Gemm/Gemv::from_device
struct ShaderDefs { bool row_major = false; };
inline ShaderDefs row_major_shader_defs() { return ShaderDefs{ true }; }

// gemm.rs:9-127
struct Gemm {
    bool row_major = false;
    static Gemm from_device(const Device &, ShaderDefs defs = {}) { return Gemm{ defs.row_major }; } // pipelines are ahead-of-time compiled
    template <typename T>
    void dispatch_generic(const Device &, const ViewShapeBuffers &, ComputePass &pass, GpuTensorView<T> out, GpuTensorView<T> m1,
                          GpuTensorView<T> m2, GemmVariant variant) const {
        check((row_major ? wg_gemm_rm : wg_gemm)(pass.ctx(), (wg_gemm_variant)variant, dtype_of<T>::value, out.buffer(), out.shape(),
                                                 m1.buffer(), m1.shape(), m2.buffer(), m2.shape()));
    }
    template <typename T>
    void dispatch(const Device &d, const ViewShapeBuffers &s, ComputePass &p, GpuTensorView<T> out, GpuTensorView<T> m1, GpuTensorView<T> m2) const {
        dispatch_generic(d, s, p, out, m1, m2, GemmVariant::Gemm);
    }
    template <typename T> // extension: out = alpha * op(m1) * m2 + beta * out
    void dispatch_ex(const Device &, const ViewShapeBuffers &, ComputePass &pass, float alpha, float beta, GpuTensorView<T> out,
                     GpuTensorView<T> m1, GpuTensorView<T> m2, GemmVariant variant = GemmVariant::Gemm) const {
        check(wg_gemm_ex(pass.ctx(), (wg_gemm_variant)variant, dtype_of<T>::value, alpha, beta, out.buffer(), out.shape(), m1.buffer(),
                         m1.shape(), m2.buffer(), m2.shape()));
    }
    template <typename T>
    void dispatch_tr(const Device &d, const ViewShapeBuffers &s, ComputePass &p, GpuTensorView<T> out, GpuTensorView<T> m1, GpuTensorView<T> m2) const {
        dispatch_generic(d, s, p, out, m1, m2, GemmVariant::GemmTr);
    }
};

// gemv.rs:9-137
struct Gemv {
    bool row_major = false;
    static Gemv from_device(const Device &, ShaderDefs defs = {}) { return Gemv{ defs.row_major }; }
    template <typename T>
    void dispatch_generic(const Device &, const ViewShapeBuffers &, ComputePass &pass, GpuTensorView<T> out, GpuTensorView<T> m,
                          GpuTensorView<T> v, GemvVariant variant) const {
        check((row_major ? wg_gemv_rm : wg_gemv)(pass.ctx(), (wg_gemv_variant)variant, dtype_of<T>::value, out.buffer(), out.shape(),
                                                 m.buffer(), m.shape(), v.buffer(), v.shape()));
    }
    template <typename T>
    void dispatch(const Device &d, const ViewShapeBuffers &s, ComputePass &p, GpuTensorView<T> out, GpuTensorView<T> m, GpuTensorView<T> v) const {
        dispatch_generic(d, s, p, out, m, v, GemvVariant::Gemv);
    }
    template <typename T>
    void dispatch_tr(const Device &d, const ViewShapeBuffers &s, ComputePass &p, GpuTensorView<T> out, GpuTensorView<T> m, GpuTensorView<T> v) const {
        dispatch_generic(d, s, p, out, m, v, GemvVariant::GemvTr);
    }
};

// reduce.rs:62-113
struct Reduce {
    ReduceOp op;
    static Reduce create(const Device &, ReduceOp op) { return {op}; } // Reduce::new(device, op)
    template <typename T>
    void dispatch(const Device &, const ViewShapeBuffers &, ComputePass &pass, GpuTensorView<T> value, const GpuScalar<T> &result) const {
        check(wg_reduce(pass.ctx(), (wg_reduce_op)op, dtype_of<T>::value, value.buffer(), value.shape(), result.buffer()));
    }
    template <typename T> // extension: two-pass multi-workgroup reduce of one long vector (Min/Max same bits; Sum/Prod/SqNorm re-associated)
    void dispatch_fast(const Device &, const ViewShapeBuffers &, ComputePass &pass, GpuTensorView<T> value, const GpuScalar<T> &result) const {
        check(wg_reduce_fast(pass.ctx(), (wg_reduce_op)op, dtype_of<T>::value, value.buffer(), value.shape(), result.buffer()));
    }
    template <typename T> // extension: every column of a matrix/cube view in one launch
    void dispatch_batched(const Device &, const ViewShapeBuffers &, ComputePass &pass, GpuTensorView<T> values, const GpuVector<T> &results) const {
        check(wg_reduce_batched(pass.ctx(), (wg_reduce_op)op, dtype_of<T>::value, values.buffer(), values.shape(), results.buffer()));
    }
};

// op_assign.rs:43-95
struct OpAssign {
    OpAssignVariant op;
    static OpAssign create(const Device &, OpAssignVariant op) { return {op}; } // OpAssign::new(device, op)
    template <typename T>
    void dispatch(const Device &, const ViewShapeBuffers &, ComputePass &pass, GpuTensorView<T> in_out_a, GpuTensorView<T> in_b) const {
        check(wg_op_assign(pass.ctx(), (wg_op_assign_variant)op, dtype_of<T>::value, in_out_a.buffer(), in_out_a.shape(), in_b.buffer(),
                           in_b.shape()));
    }
};

// extension: y = fma(alpha, x, y) (the north-star's "Axpy"; the reference has only OpAssign)
struct Axpy {
    static Axpy from_device(const Device &) { return {}; }
    template <typename T>
    void dispatch(const Device &, const ViewShapeBuffers &, ComputePass &pass, float alpha, GpuTensorView<T> in_out_y, GpuTensorView<T> in_x) const {
        check(wg_axpy(pass.ctx(), alpha, dtype_of<T>::value, in_out_y.buffer(), in_out_y.shape(), in_x.buffer(), in_x.shape()));
    }
};

// extension: dst view = src view where it has elements, 0 elsewhere -- any offset, stride and length on either side (wg_copy_view: the aligned copy of an odd view, made once)
struct CopyView {
    static CopyView from_device(const Device &) { return {}; }
    template <typename T>
    void dispatch(const Device &, const ViewShapeBuffers &, ComputePass &pass, GpuTensorView<T> dst, GpuTensorView<T> src) const {
        check(wg_copy_view(pass.ctx(), dtype_of<T>::value, dst.buffer(), dst.shape(), src.buffer(), src.shape()));
    }
};

} // namespace wgebra
