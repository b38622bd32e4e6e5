// drt_pybind.cpp -- thin pybind11 shim over the C ABI (include/drt_hip.h).
//
// Pointers cross as integers (torch.Tensor.data_ptr()); the GIL is released around
// every call that enqueues or waits for device work; a non-zero status becomes
// RuntimeError(drt_last_error()).  No torch headers, no logic.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/drt_hip.h"

namespace py = pybind11;

namespace {

template <typename T>
T *ptr(uintptr_t p) { return reinterpret_cast<T *>(p); }

class Integrator {
public:
    Integrator(const py::dict &props, int device)
    {
        drt_config c{};
        auto geti = [&](const char *k, int dflt) {
            return props.contains(k) ? py::cast<int>(py::int_(py::cast<py::object>(props[k]))) : dflt;
        };
        auto getb = [&](const char *k, bool dflt) {
            return props.contains(k) ? (py::cast<bool>(props[k]) ? 1 : 0) : (dflt ? 1 : 0);
        };
        if (!props.contains("max_depth")) throw std::invalid_argument("props must contain max_depth");
        c.hide_emitters = getb("hide_emitters", false);
        c.use_nee = getb("use_nee", true);
        c.use_drt = getb("use_drt", true);
        c.use_drt_subsampling = getb("use_drt_subsampling", true);
        c.use_drt_mis = getb("use_drt_mis", true);
        c.max_depth = geti("max_depth", 0);
        c.rr_depth = geti("rr_depth", c.max_depth + 1000);
        int rc = drt_create(&c, device, &h_);
        if (rc) throw std::runtime_error(std::string("drt_create: ") + drt_last_error(nullptr));
    }
    ~Integrator() { drt_destroy(h_); }
    Integrator(const Integrator &) = delete;
    Integrator &operator=(const Integrator &) = delete;

    void check(int rc, const char *what) const
    {
        if (rc) throw std::runtime_error(std::string(what) + ": " + drt_last_error(h_));
    }

    void release_scratch()
    {
        int rc;
        { py::gil_scoped_release nogil; rc = drt_release_scratch(h_); }
        check(rc, "drt_release_scratch");
    }
    void set_stream(uintptr_t s) { check(drt_set_stream(h_, ptr<void>(s)), "drt_set_stream"); }
    void synchronize()
    {
        py::gil_scoped_release nogil;
        int rc = drt_synchronize(h_);
        py::gil_scoped_acquire gil;
        check(rc, "drt_synchronize");
    }
    void set_ray_interleave(uint64_t chunk, uint64_t stride)
    {
        check(drt_set_ray_interleave(h_, chunk, stride), "drt_set_ray_interleave");
    }
    void set_medium(uintptr_t sigma_t, uintptr_t albedo, std::array<int32_t, 3> res,
                    std::array<float, 3> bmin, std::array<float, 3> bmax, float scale, int factor)
    {
        int rc;
        {
            py::gil_scoped_release nogil;
            rc = drt_set_medium(h_, ptr<const float>(sigma_t), ptr<const float>(albedo), res.data(),
                                bmin.data(), bmax.data(), scale, factor);
        }
        check(rc, "drt_set_medium");
    }
    uint64_t nerf_tile_lds_adds() { uint64_t v = 0; check(drt_nerf_tile_stats(h_, &v), "drt_nerf_tile_stats"); return v; }
    void set_colour_resolution(std::array<int32_t, 3> res) { check(drt_set_colour_resolution(h_, res.data()), "drt_set_colour_resolution"); }
    void params_changed()
    {
        int rc;
        { py::gil_scoped_release nogil; rc = drt_params_changed(h_); }
        check(rc, "drt_params_changed");
    }
    void set_emitter_constant(std::array<float, 3> rgb)
    {
        check(drt_set_emitter_constant(h_, rgb.data()), "drt_set_emitter_constant");
    }
    void set_emitter_envmap(uintptr_t pixels, int w, int hgt, std::array<float, 9> to_world, float scale)
    {
        int rc;
        {
            py::gil_scoped_release nogil;
            rc = drt_set_emitter_envmap(h_, (const float *) pixels, w, hgt, to_world.data(), scale);
        }
        check(rc, "drt_set_emitter_envmap");
    }
    void set_sensor_perspective(std::array<float, 3> o, std::array<float, 3> left, std::array<float, 3> up,
                                std::array<float, 3> dir, float tan_x, float tan_y, int w, int hgt)
    {
        check(drt_set_sensor_perspective(h_, o.data(), left.data(), up.data(), dir.data(), tan_x, tan_y, w, hgt),
              "drt_set_sensor_perspective");
    }
    void render_primal(uintptr_t rays_o, uintptr_t rays_d, uint64_t n, uint64_t off, uint32_t spp,
                       uint32_t seed, uintptr_t L_out)
    {
        int rc;
        {
            py::gil_scoped_release nogil;
            rc = drt_render_primal(h_, ptr<const float>(rays_o), ptr<const float>(rays_d), n, off, spp, seed,
                                   ptr<float>(L_out));
        }
        check(rc, "drt_render_primal");
    }
    void render_backward(uintptr_t rays_o, uintptr_t rays_d, uint64_t n, uint64_t off, uint32_t spp,
                         uint32_t seed, uintptr_t dL, uintptr_t L_in, uintptr_t g_sigma, uintptr_t g_albedo)
    {
        int rc;
        {
            py::gil_scoped_release nogil;
            rc = drt_render_backward(h_, ptr<const float>(rays_o), ptr<const float>(rays_d), n, off, spp, seed,
                                     ptr<const float>(dL), ptr<const float>(L_in), ptr<float>(g_sigma),
                                     ptr<float>(g_albedo));
        }
        check(rc, "drt_render_backward");
    }
    static drt_nerf_config nerf_cfg(const py::dict &p)
    {
        drt_nerf_config c{};
        c.hide_emitters = p.contains("hide_emitters") && py::cast<bool>(p["hide_emitters"]);
        c.queries_per_ray = p.contains("queries_per_ray") ? py::cast<int>(p["queries_per_ray"]) : 128;
        c.jittering_enabled = p.contains("jittering_enabled") ? (py::cast<bool>(p["jittering_enabled"]) ? 1 : 0) : 1;
        c.activation_relu = p.contains("activation_relu") && py::cast<bool>(p["activation_relu"]);
        return c;
    }
    void nerf_render_primal(const py::dict &props, uintptr_t emission, uintptr_t rays_o, uintptr_t rays_d, uint64_t n,
                            uint64_t off, uint32_t spp, uint32_t seed, uintptr_t L_out)
    {
        drt_nerf_config c = nerf_cfg(props);
        int rc;
        {
            py::gil_scoped_release nogil;
            rc = drt_nerf_render_primal(h_, &c, ptr<const float>(emission), ptr<const float>(rays_o),
                                        ptr<const float>(rays_d), n, off, spp, seed, ptr<float>(L_out));
        }
        check(rc, "drt_nerf_render_primal");
    }
    void nerf_render_backward(const py::dict &props, uintptr_t emission, uintptr_t rays_o, uintptr_t rays_d, uint64_t n,
                              uint64_t off, uint32_t spp, uint32_t seed, uintptr_t dL, uintptr_t L_in,
                              uintptr_t g_sigma, uintptr_t g_emission)
    {
        drt_nerf_config c = nerf_cfg(props);
        int rc;
        {
            py::gil_scoped_release nogil;
            rc = drt_nerf_render_backward(h_, &c, ptr<const float>(emission), ptr<const float>(rays_o),
                                          ptr<const float>(rays_d), n, off, spp, seed, ptr<const float>(dL),
                                          ptr<const float>(L_in), ptr<float>(g_sigma), ptr<float>(g_emission));
        }
        check(rc, "drt_nerf_render_backward");
    }
    void fused_render_primal(const py::dict &props, uintptr_t rays_o, uintptr_t rays_d, uint64_t n, uint64_t off, uint32_t spp,
                             uint32_t seed, uintptr_t L_nerf, uintptr_t L_drt)
    {
        drt_nerf_config c = nerf_cfg(props);
        int rc;
        {
            py::gil_scoped_release nogil;
            rc = drt_fused_render_primal(h_, &c, ptr<const float>(rays_o), ptr<const float>(rays_d), n, off, spp, seed,
                                         ptr<float>(L_nerf), ptr<float>(L_drt));
        }
        check(rc, "drt_fused_render_primal");
    }
    void fused_render_backward(const py::dict &props, uintptr_t rays_o, uintptr_t rays_d, uint64_t n, uint64_t off, uint32_t spp,
                               uint32_t seed, uintptr_t dL_nerf, uintptr_t L_nerf_in, uintptr_t dL_drt, uintptr_t L_drt_in,
                               uintptr_t g_sigma, uintptr_t g_rgb)
    {
        drt_nerf_config c = nerf_cfg(props);
        int rc;
        {
            py::gil_scoped_release nogil;
            rc = drt_fused_render_backward(h_, &c, ptr<const float>(rays_o), ptr<const float>(rays_d), n, off, spp, seed,
                                           ptr<const float>(dL_nerf), ptr<const float>(L_nerf_in), ptr<const float>(dL_drt),
                                           ptr<const float>(L_drt_in), ptr<float>(g_sigma), ptr<float>(g_rgb));
        }
        check(rc, "drt_fused_render_backward");
    }
    void batch_sample_rays(uintptr_t sensors, int n_sensors, uint32_t batch, uint32_t spp, uint32_t seed_px, uint32_t seed_rays,
                           uintptr_t ro, uintptr_t rd, uintptr_t sidx, uintptr_t pix, uint32_t batch_first)
    {
        int rc;
        {
            py::gil_scoped_release nogil;
            rc = drt_batch_sample_rays_range(h_, ptr<const float>(sensors), n_sensors, batch_first, batch, spp, seed_px, seed_rays,
                                             ptr<float>(ro), ptr<float>(rd), ptr<uint32_t>(sidx), ptr<uint32_t>(pix));
        }
        check(rc, "drt_batch_sample_rays_range");
    }
    void film_develop(uintptr_t L, uint64_t n_pixels, uint32_t spp, uintptr_t image)
    {
        int rc;
        { py::gil_scoped_release nogil; rc = drt_film_develop(h_, ptr<const float>(L), n_pixels, spp, ptr<float>(image)); }
        check(rc, "drt_film_develop");
    }
    void film_backward(uintptr_t grad_image, uint64_t n_pixels, uint32_t spp, uintptr_t dL)
    {
        int rc;
        { py::gil_scoped_release nogil; rc = drt_film_backward(h_, ptr<const float>(grad_image), n_pixels, spp, ptr<float>(dL)); }
        check(rc, "drt_film_backward");
    }
    void debug_eval(int op, uintptr_t in, uint64_t n, uintptr_t out)
    {
        int rc;
        { py::gil_scoped_release nogil; rc = drt_debug_eval(h_, op, ptr<const float>(in), n, ptr<float>(out)); }
        check(rc, "drt_debug_eval");
    }
    void set_debug_flags(uint32_t f) { check(drt_set_debug_flags(h_, f), "drt_set_debug_flags"); }
    void enable_counters(bool on) { check(drt_enable_counters(h_, on ? 1 : 0), "drt_enable_counters"); }
    void reset_counters() { check(drt_reset_counters(h_), "drt_reset_counters"); }
    py::dict get_counters()
    {
        drt_counters c{};
        int rc;
        { py::gil_scoped_release nogil; rc = drt_get_counters(h_, &c); }
        check(rc, "drt_get_counters");
        py::dict d;
        d["n_rays"] = c.n_rays; d["n_dt"] = c.n_dt; d["n_rt"] = c.n_rt; d["n_drt"] = c.n_drt;
        d["n_alb"] = c.n_alb; d["n_tr"] = c.n_tr; d["n_rt_adj"] = c.n_rt_adj; d["n_sc"] = c.n_sc;
        d["n_sc_alb"] = c.n_sc_alb;
        return d;
    }
    void enable_timing(bool on) { check(drt_enable_timing(h_, on ? 1 : 0), "drt_enable_timing"); }
    std::vector<float> read_timings(int kind)
    {
        int n = drt_read_timings(h_, kind, nullptr, 0);
        if (n < 0) check(n, "drt_read_timings");
        std::vector<float> out((size_t) n);
        if (n > 0) {
            int rc = drt_read_timings(h_, kind, out.data(), n);
            if (rc < 0) check(rc, "drt_read_timings");
        }
        return out;
    }

private:
    drt_handle h_ = nullptr;
};

}  // namespace

#ifndef DRT_PYBIND_NAME
#define DRT_PYBIND_NAME _drt_pybind
#endif
PYBIND11_MODULE(DRT_PYBIND_NAME, m)
{
    m.doc() = "pybind11 shim over libdrt_hip.so (C ABI in include/drt_hip.h)";
    m.def("version", []() { return std::string(drt_version()); });
    m.def("adam_step", [](uintptr_t stream, uintptr_t p, uintptr_t g, uintptr_t mm, uintptr_t v, uint64_t n, double b1, double b2, double eps, double lr_t) {
        const int rc = drt_adam_step(reinterpret_cast<void *>(stream), reinterpret_cast<float *>(p), reinterpret_cast<const float *>(g),
                                     reinterpret_cast<float *>(mm), reinterpret_cast<float *>(v), n, b1, b2, eps, lr_t);
        if (rc != DRT_OK) throw std::runtime_error("drt_adam_step failed (code " + std::to_string(rc) + ")");
    });
    m.def("adam_step_clamped", [](uintptr_t stream, uintptr_t p, uintptr_t g, uintptr_t mm, uintptr_t v, uint64_t n, double b1, double b2, double eps, double lr_t, float lo, float hi) {
        const int rc = drt_adam_step_clamped(reinterpret_cast<void *>(stream), reinterpret_cast<float *>(p), reinterpret_cast<const float *>(g),
                                     reinterpret_cast<float *>(mm), reinterpret_cast<float *>(v), n, b1, b2, eps, lr_t, lo, hi);
        if (rc != DRT_OK) throw std::runtime_error("drt_adam_step_clamped failed (code " + std::to_string(rc) + ")");
    });
    m.def("grad_support_mask", [](uintptr_t stream, uintptr_t sigma_t, int rx, int ry, int rz, uint64_t sparse_off, uint32_t channels,
                                  uint64_t n_blocks, uint32_t block_floats, uintptr_t bits, uintptr_t mask) {
        const int32_t res[3] = { rx, ry, rz };
        const int rc = drt_grad_support_mask(reinterpret_cast<void *>(stream), reinterpret_cast<const float *>(sigma_t), res, sparse_off, channels,
                                             n_blocks, block_floats, reinterpret_cast<uint32_t *>(bits), reinterpret_cast<uint8_t *>(mask));
        if (rc != DRT_OK) throw std::runtime_error("drt_grad_support_mask failed (code " + std::to_string(rc) + ")");
    });
    m.def("grad_block_mask", [](uintptr_t stream, uintptr_t buf, uint64_t n_blocks, uint32_t block_floats, uintptr_t mask) {
        const int rc = drt_grad_block_mask(reinterpret_cast<void *>(stream), reinterpret_cast<const float *>(buf), n_blocks,
                                           block_floats, reinterpret_cast<uint8_t *>(mask));
        if (rc != DRT_OK) throw std::runtime_error("drt_grad_block_mask failed (code " + std::to_string(rc) + ")");
    });
    m.def("grad_block_positions", [](uintptr_t stream, uintptr_t mask, uint64_t n_blocks, uintptr_t pos, uintptr_t count, uintptr_t scratch) {
        const int rc = drt_grad_block_positions(reinterpret_cast<void *>(stream), reinterpret_cast<const uint8_t *>(mask), n_blocks,
                                                reinterpret_cast<int32_t *>(pos), reinterpret_cast<int32_t *>(count), reinterpret_cast<uint32_t *>(scratch));
        if (rc != DRT_OK) throw std::runtime_error("drt_grad_block_positions failed (code " + std::to_string(rc) + ")");
    });
    m.def("grad_pack", [](uintptr_t stream, uintptr_t flat, uintptr_t pos, uint64_t n_blocks, uint32_t block_floats, uintptr_t packed, uintptr_t check) {
        const int rc = drt_grad_pack(reinterpret_cast<void *>(stream), reinterpret_cast<const float *>(flat), reinterpret_cast<const int32_t *>(pos), n_blocks,
                                     block_floats, reinterpret_cast<float *>(packed), reinterpret_cast<float *>(check));
        if (rc != DRT_OK) throw std::runtime_error("drt_grad_pack failed (code " + std::to_string(rc) + ")");
    });
    m.def("grad_unpack", [](uintptr_t stream, uintptr_t packed, uintptr_t pos, uint64_t n_blocks, uint32_t block_floats, uintptr_t flat) {
        const int rc = drt_grad_unpack(reinterpret_cast<void *>(stream), reinterpret_cast<const float *>(packed), reinterpret_cast<const int32_t *>(pos), n_blocks,
                                       block_floats, reinterpret_cast<float *>(flat));
        if (rc != DRT_OK) throw std::runtime_error("drt_grad_unpack failed (code " + std::to_string(rc) + ")");
    });
    py::class_<Integrator>(m, "Integrator", py::module_local())   // (two flavours of this module can live in one process)
        .def(py::init<const py::dict &, int>(), py::arg("props"), py::arg("device") = 0)
        .def("set_stream", &Integrator::set_stream)
        .def("synchronize", &Integrator::synchronize)
        .def("release_scratch", &Integrator::release_scratch)
        .def("set_ray_interleave", &Integrator::set_ray_interleave)
        .def("set_medium", &Integrator::set_medium)
        .def("set_colour_resolution", &Integrator::set_colour_resolution)
        .def("nerf_tile_lds_adds", &Integrator::nerf_tile_lds_adds)
        .def("params_changed", &Integrator::params_changed)
        .def("set_emitter_constant", &Integrator::set_emitter_constant)
        .def("set_emitter_envmap", &Integrator::set_emitter_envmap)
        .def("set_sensor_perspective", &Integrator::set_sensor_perspective)
        .def("render_primal", &Integrator::render_primal)
        .def("render_backward", &Integrator::render_backward)
        .def("nerf_render_primal", &Integrator::nerf_render_primal)
        .def("nerf_render_backward", &Integrator::nerf_render_backward)
        .def("fused_render_primal", &Integrator::fused_render_primal)
        .def("fused_render_backward", &Integrator::fused_render_backward)
        .def("batch_sample_rays", &Integrator::batch_sample_rays, py::arg("sensors"), py::arg("n_sensors"), py::arg("batch"),
             py::arg("spp"), py::arg("seed_px"), py::arg("seed_rays"), py::arg("ro"), py::arg("rd"), py::arg("sidx"),
             py::arg("pix"), py::arg("batch_first") = 0)
        .def("film_develop", &Integrator::film_develop)
        .def("film_backward", &Integrator::film_backward)
        .def("debug_eval", &Integrator::debug_eval)
        .def("set_debug_flags", &Integrator::set_debug_flags)
        .def("enable_counters", &Integrator::enable_counters)
        .def("reset_counters", &Integrator::reset_counters)
        .def("get_counters", &Integrator::get_counters)
        .def("enable_timing", &Integrator::enable_timing)
        .def("read_timings", &Integrator::read_timings);
}
