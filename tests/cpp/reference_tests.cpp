// The reference's four hot-path tests (gemm.rs:141-202, gemv.rs:152-197, reduce.rs:136-179, op_assign.rs:108-157)
// written against include/wgebra.hpp -- the compiled-language host side over the C ABI.  The CPU check is what the
// reference uses (a plain nalgebra-style product / fold), in double precision here, with the reference's epsilons.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "wgebra.hpp"

using namespace wgebra;

static int failures = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

static std::vector<float> new_random(size_t n, uint32_t seed) { // DMatrix::<f32>::new_random: U[0,1)
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> d(0.f, 1.f);
    std::vector<float> v(n);
    for (auto &x : v) x = d(rng);
    return v;
}

static void gpu_gemm(const GpuInstance &gpu) {
    auto gemm = Gemm::from_device(gpu.device());
    auto shapes = ViewShapeBuffers::create();
    const uint32_t NROWS = 256, NCOLS = 256;
    auto m1_cpu = new_random(NROWS * NCOLS, 1), m2_cpu = new_random(NCOLS * NROWS, 2);
    std::vector<float> lhs_cpu(NROWS * NROWS, 0.f);
    auto m1 = TensorBuilder::matrix(NROWS, NCOLS, BufferUsages::STORAGE).build_init(gpu.device(), m1_cpu);
    auto m2 = TensorBuilder::matrix(NCOLS, NROWS, BufferUsages::STORAGE).build_init(gpu.device(), m2_cpu);
    auto result = TensorBuilder::matrix(NROWS, NROWS, BufferUsages::STORAGE | BufferUsages::COPY_SRC).build_init(gpu.device(), lhs_cpu);
    auto staging = TensorBuilder::matrix(NROWS, NROWS, BufferUsages::MAP_READ | BufferUsages::COPY_DST).build<float>(gpu.device());
    for (auto variant : {GemmVariant::Gemm, GemmVariant::GemmTr, GemmVariant::GemmFast, GemmVariant::GemmTrFast}) {
        auto encoder = gpu.create_command_encoder();
        auto pass = encoder.compute_pass("test", nullptr);
        gemm.dispatch_generic<float>(gpu.device(), shapes, pass, result.as_embedded_view(), m1.as_embedded_view(), m2.as_embedded_view(), variant);
        staging.copy_from(encoder, result);
        gpu.queue().submit(encoder.finish());
        auto gpu_result = staging.read(gpu.device());
        const bool tr = variant == GemmVariant::GemmTr || variant == GemmVariant::GemmTrFast;
        double worst = 0;
        for (uint32_t j = 0; j < NROWS; ++j)
            for (uint32_t i = 0; i < NROWS; ++i) {
                double acc = 0;
                for (uint32_t k = 0; k < NCOLS; ++k)
                    acc += (double)(tr ? m1_cpu[k + i * NROWS] : m1_cpu[i + k * NROWS]) * m2_cpu[k + j * NCOLS];
                worst = std::fmax(worst, std::fabs(acc - gpu_result[i + j * NROWS]));
            }
        EXPECT(worst <= 1.0e-3, "gpu_gemm variant %d: max abs err %g", (int)variant, worst);
    }
}

static void gpu_gemv(const GpuInstance &gpu) {
    auto gemv = Gemv::from_device(gpu.device());
    auto shapes = ViewShapeBuffers::create();
    const uint32_t NROWS = 1024, NCOLS = 1024;
    auto m_cpu = new_random(NROWS * NCOLS, 3), v_cpu = new_random(NCOLS, 4), lhs_cpu = new_random(NROWS, 5);
    auto m = TensorBuilder::matrix(NROWS, NCOLS, BufferUsages::STORAGE).build_init(gpu.device(), m_cpu);
    auto v = TensorBuilder::vector(NCOLS, BufferUsages::STORAGE).build_init(gpu.device(), v_cpu);
    auto result = TensorBuilder::vector(NROWS, BufferUsages::STORAGE | BufferUsages::COPY_SRC).build_init(gpu.device(), lhs_cpu);
    auto staging = TensorBuilder::vector(NROWS, BufferUsages::MAP_READ | BufferUsages::COPY_DST).build<float>(gpu.device());
    for (auto variant : {GemvVariant::Gemv, GemvVariant::GemvTr, GemvVariant::GemvFast, GemvVariant::GemvTrFast}) {
        auto encoder = gpu.create_command_encoder();
        auto pass = encoder.compute_pass("test", nullptr);
        gemv.dispatch_generic<float>(gpu.device(), shapes, pass, result, m, v, variant);
        staging.copy_from(encoder, result);
        gpu.queue().submit(encoder.finish());
        auto gpu_result = staging.read(gpu.device());
        const bool tr = variant == GemvVariant::GemvTr || variant == GemvVariant::GemvTrFast;
        double worst = 0;
        for (uint32_t i = 0; i < NROWS; ++i) {
            double acc = 0;
            for (uint32_t k = 0; k < NCOLS; ++k) acc += (double)(tr ? m_cpu[k + i * NROWS] : m_cpu[i + k * NROWS]) * v_cpu[k];
            worst = std::fmax(worst, std::fabs(acc - gpu_result[i]));
        }
        EXPECT(worst <= 1.0e-3, "gpu_gemv variant %d: max abs err %g", (int)variant, worst);
    }
}

static void gpu_reduce(const GpuInstance &gpu) {
    auto shapes = ViewShapeBuffers::create();
    for (auto op : {ReduceOp::Min, ReduceOp::Max, ReduceOp::Sum, ReduceOp::SqNorm, ReduceOp::Prod}) {
        auto reduce = Reduce::create(gpu.device(), op);
        const uint32_t LEN = 345;
        auto numbers = new_random(LEN, 6 + (int)op);
        auto vector = TensorBuilder::vector(LEN, BufferUsages::STORAGE).build_init(gpu.device(), numbers);
        auto result = TensorBuilder::scalar(BufferUsages::STORAGE | BufferUsages::COPY_SRC).build<float>(gpu.device());
        auto staging = TensorBuilder::scalar(BufferUsages::MAP_READ | BufferUsages::COPY_DST).build<float>(gpu.device());
        auto encoder = gpu.create_command_encoder();
        auto pass = encoder.compute_pass("test", nullptr);
        reduce.dispatch<float>(gpu.device(), shapes, pass, vector, result);
        staging.copy_from(encoder, result);
        gpu.queue().submit(encoder.finish());
        double expect = op == ReduceOp::Min ? 1e30 : op == ReduceOp::Max ? -1e30 : op == ReduceOp::Prod ? 1.0 : 0.0;
        for (float x : numbers) {
            if (op == ReduceOp::Min) expect = std::fmin(expect, x);
            else if (op == ReduceOp::Max) expect = std::fmax(expect, x);
            else if (op == ReduceOp::Sum) expect += x;
            else if (op == ReduceOp::SqNorm) expect += (double)x * x;
            else expect *= x;
        }
        const double got = staging.read(gpu.device())[0];
        EXPECT(std::fabs(got - expect) <= 1.0e-3, "gpu_reduce op %d: got %g expected %g", (int)op, got, expect);
    }
}

static void gpu_op_assign(const GpuInstance &gpu) {
    auto shapes = ViewShapeBuffers::create();
    for (auto op : {OpAssignVariant::Add, OpAssignVariant::Sub, OpAssignVariant::Mul, OpAssignVariant::Div}) {
        auto op_assign = OpAssign::create(gpu.device(), op);
        const uint32_t LEN = 1757;
        std::vector<float> v0(LEN), v1(LEN);
        for (uint32_t i = 0; i < LEN; ++i) { v0[i] = (float)i + 0.1f; v1[i] = (float)i * 10.0f + 0.1f; }
        auto gpu_v0 = TensorBuilder::vector(LEN, BufferUsages::STORAGE | BufferUsages::COPY_SRC).build_init(gpu.device(), v0);
        auto gpu_v1 = TensorBuilder::vector(LEN, BufferUsages::STORAGE).build_init(gpu.device(), v1);
        auto staging = TensorBuilder::vector(LEN, BufferUsages::MAP_READ | BufferUsages::COPY_DST).build<float>(gpu.device());
        auto encoder = gpu.create_command_encoder();
        auto pass = encoder.compute_pass("test", nullptr);
        op_assign.dispatch<float>(gpu.device(), shapes, pass, gpu_v0, gpu_v1);
        staging.copy_from(encoder, gpu_v0);
        gpu.queue().submit(encoder.finish());
        auto got = staging.read(gpu.device());
        for (uint32_t i = 0; i < LEN; ++i) {
            volatile float e = op == OpAssignVariant::Add ? v0[i] + v1[i] : op == OpAssignVariant::Sub ? v0[i] - v1[i]
                             : op == OpAssignVariant::Mul ? v0[i] * v1[i] : v0[i] / v1[i];
            if (got[i] != e) { EXPECT(false, "gpu_op_assign op %d: element %u got %.9g expected %.9g (must be bit-exact)", (int)op, i, got[i], (float)e); break; }
        }
    }
}

static void panics(const GpuInstance &gpu) {
    auto shapes = ViewShapeBuffers::create();
    auto a = TensorBuilder::matrix(8, 12, BufferUsages::STORAGE).build<float>(gpu.device());
    auto b = TensorBuilder::matrix(8, 8, BufferUsages::STORAGE).build<float>(gpu.device());
    auto encoder = gpu.create_command_encoder();
    auto pass = encoder.compute_pass("test", nullptr);
    bool threw = false;
    try { Gemm::from_device(gpu.device()).dispatch<float>(gpu.device(), shapes, pass, b, a, b); }
    catch (const wgcore::Panic &e) { threw = std::string(e.what()).find("Gemm: dimension mismatch.") == 0; }
    EXPECT(threw, "dimension mismatch must panic with the reference's message");
}

// A tensor and a recorded command buffer that outlive the GpuInstance they were created on: their deleters (wg_buf_destroy,
// wg_cmdbuf_destroy) dereference the context, so they share its ownership -- like wgpu buffers keep their Device alive.
static void handles_outlive_the_instance() {
    std::vector<float> data(1024, 2.5f);
    auto make = [&]() {
        auto gpu = GpuInstance::create();
        auto t = TensorBuilder::vector(1024, BufferUsages::STORAGE | BufferUsages::COPY_SRC).build_init(gpu.device(), data);
        auto enc = gpu.create_command_encoder(/*record=*/true);
        auto pass = enc.compute_pass("rec", nullptr);
        OpAssign::create(gpu.device(), OpAssignVariant::Add).dispatch<float>(gpu.device(), ViewShapeBuffers::create(), pass, t, t);
        auto cb = enc.finish();
        gpu.queue().submit(cb);
        gpu.poll_wait();
        return std::make_pair(std::move(t), std::move(cb));
    };
    auto kept = make(); // the GpuInstance is gone; the context lives on through the tensor and the command buffer
    EXPECT(kept.first.len() == 1024, "tensor survived");
} // both destroyed here, then the context

// The rest of tensor.rs in the C++ mirror: reshape / slice / as_view, copy_from_view, build_bytes / build_encase / build_uninit_encased, uninit / init / encase,
// into_inner -- and CopyView (wg_copy_view) on a view at an odd offset.
static void tensor_members(const GpuInstance &gpu) {
    const uint32_t R = 12, C = 5;
    std::vector<float> host(R * C);
    for (size_t i = 0; i < host.size(); ++i) host[i] = (float)i;
    auto m = GpuTensor<float>::init(gpu.device(), { R, C }, host, BufferUsages::STORAGE | BufferUsages::COPY_SRC);
    EXPECT(!m.is_empty() && m.bytes_len_encased() == R * C * 4, "init / is_empty / bytes_len_encased");
    auto sl = m.slice(1, 2, 4, 3); // the reference's offset: i + j * (the SLICE's nrows) = 1 + 2 * 4
    EXPECT(sl.shape().offset == 9 && sl.shape().stride == R && sl.shape().size[0] == 4 && sl.shape().size[1] == 3, "slice: the reference's offset arithmetic");
    auto rs = m.reshape({ 6, 4 });
    EXPECT(rs.shape().stride == 6 && rs.shape().stride_mat == 24 && rs.shape().offset == 0, "reshape: column-major defaults");
    // column 2 of m into a vector, through copy_from_view
    auto col = TensorBuilder::vector(R, BufferUsages::STORAGE | BufferUsages::COPY_SRC | BufferUsages::COPY_DST).build_uninit_encased<float>(gpu.device());
    auto enc = gpu.create_command_encoder();
    col.copy_from_view(enc, m.as_view().columns(2, 1));
    gpu.queue().submit(enc.finish());
    auto got = col.read(gpu.device());
    bool ok = true;
    for (uint32_t r = 0; r < R; ++r) ok &= got[r] == host[2 * R + r];
    EXPECT(ok, "copy_from_view: a column");
    // CopyView: rows 1 .. 9 of columns 1 .. 3 (offset 13: odd) into a dense 12 x 4 block -- zero beyond the source
    auto dst = GpuTensor<float>::uninit(gpu.device(), { 12, 4 }, BufferUsages::STORAGE | BufferUsages::COPY_SRC);
    auto e2 = gpu.create_command_encoder();
    auto pass = e2.compute_pass("copy", nullptr);
    wgebra::CopyView::from_device(gpu.device()).dispatch<float>(gpu.device(), ViewShapeBuffers::create(), pass, dst, m.as_view().columns(1, 3).rows(1, 9));
    gpu.queue().submit(e2.finish());
    auto d = dst.read(gpu.device());
    ok = true;
    for (uint32_t c = 0; c < 4; ++c)
        for (uint32_t r = 0; r < 12; ++r) ok &= d[c * 12 + r] == ((c < 3 && r < 9) ? host[(c + 1) * R + 1 + r] : 0.f);
    EXPECT(ok, "CopyView: an odd-offset block into a larger dense one, zero-filled");
    auto raw = TensorBuilder::vector(4, BufferUsages::STORAGE | BufferUsages::COPY_SRC).build_bytes<float>(gpu.device(), host.data(), 16);
    auto enc3 = TensorBuilder::vector(4, BufferUsages::STORAGE | BufferUsages::COPY_SRC).build_encase<float>(gpu.device(), host);
    EXPECT(raw.read(gpu.device()) == enc3.read(gpu.device()), "build_bytes == build_encase for scalar items");
    auto v = GpuTensor<float>::encase(gpu.device(), host, BufferUsages::STORAGE);
    EXPECT(v.len() == host.size(), "GpuVector::encase");
    auto inner = std::move(v).into_inner();
    EXPECT(inner && wg_buf_device_ptr(inner.get()) != nullptr, "into_inner keeps the allocation");
}

// timestamps.rs in the C++ mirror: a pass timed by the next two slots, an explicit write inside it, reserved slots all-or-none, raw values and their conversion
static void timestamps(const GpuInstance &gpu) {
    std::vector<float> ones(1 << 20, 1.f);
    auto a = TensorBuilder::vector(1 << 20, BufferUsages::STORAGE).build_init(gpu.device(), ones);
    auto b = TensorBuilder::vector(1 << 20, BufferUsages::STORAGE).build_init(gpu.device(), ones);
    auto ts = gpu.timestamps(6);
    EXPECT(ts.is_empty(), "timestamps: empty at first");
    auto enc = gpu.create_command_encoder();
    uint32_t mid = 99;
    {
        auto pass = enc.compute_pass("timed", ts);
        OpAssign::create(gpu.device(), OpAssignVariant::Add).dispatch<float>(gpu.device(), ViewShapeBuffers::create(), pass, a, b);
        EXPECT(ts.write_next_timestamp(pass, &mid) && mid == 2, "timestamps: the third slot, inside the pass");
        OpAssign::create(gpu.device(), OpAssignVariant::Add).dispatch<float>(gpu.device(), ViewShapeBuffers::create(), pass, a, b);
        EXPECT(!ts.write_timestamp_at(pass, 6), "timestamps: a slot past the capacity is refused");
    } // drop(pass): the end-of-pass slot
    ts.resolve(enc);
    gpu.queue().submit(enc.finish());
    uint32_t first = 0;
    EXPECT(!ts.next_query_indices(4, &first) && ts.len() == 3, "timestamps: 3 + 4 > 6: none taken");
    EXPECT(ts.next_query_indices(3, &first) && first == 3 && ts.len() == 6, "timestamps: the last three slots");
    auto ms = ts.wait_for_results_ms();
    EXPECT(ms.size() == 6 && ms[0] == 0.0 && ms[2] > 0.0 && ms[1] > ms[2] && ms[1] < 1000.0 && ms[3] == 0.0, "timestamps: begin < inside < end, unwritten slots read 0");
    auto back = GpuTimestamps::timestamps_to_ms(ts.wait_for_results(), gpu.queue().get_timestamp_period());
    EXPECT(back.size() == 6 && std::fabs(back[1] - ms[1]) < 1e-5, "timestamps: raw values x period");
    ts.clear();
    EXPECT(ts.is_empty(), "timestamps: clear");
}

int main() {
    try {
        handles_outlive_the_instance();
        auto gpu = GpuInstance::create();
        gpu_gemm(gpu);
        gpu_gemv(gpu);
        gpu_reduce(gpu);
        gpu_op_assign(gpu);
        panics(gpu);
        tensor_members(gpu);
        timestamps(gpu);
    } catch (const std::exception &e) {
        std::printf("FAIL: exception: %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "%d FAILURES\n" : "ALL OK (%d failures)\n", failures);
    return failures ? 1 : 0;
}
