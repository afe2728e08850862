// Host build of the geometry header (include/wgebra_geometry.hpp) behind the same item layouts as wg_geometry_apply, so that
// the CPU test-suite exercises the very code the HIP kernels compile (tests/test_geometry.py builds this with g++).
#include "../../wgmath_amd/csrc/geometry_items.hpp"

extern "C" int geom_apply_host(int op, unsigned dim, const float *in, float *out, unsigned count) {
    using namespace wgg_items;
    const unsigned nin = in_floats(op, dim), nout = out_floats(op, dim);
    if (nout == 0) return 1;
    for (unsigned i = 0; i < count; ++i) {
        const float *p = in + (size_t)i * nin;
        float *o = out + (size_t)i * nout;
        if (!is_mat_op(op) && op >= OP_QUAT_RAW) raw_item(op, p, o);
        else if (!is_mat_op(op)) transform_item(op, p, o);
        else if (dim == 2) mat_item<2>(op, p, o);
        else if (dim == 3) mat_item<3>(op, p, o);
        else if (dim == 4) mat_item<4>(op, p, o);
        else return 2;
    }
    return 0;
}
