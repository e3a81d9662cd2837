// python_api.cpp — the `pyngp` module: same module / class / method / property names as the reference's binding
// (src/python_api.cu:306-888) for the NeRF train + render path, so scripts/run.py-style drivers run unchanged.
// Long calls release the GIL like the reference (python_api.cu:546, 566, 594).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "testbed.h"
#include "snapshot.h"
#include "nerf_renderer.h"
#include "nerf_loader.h"
#include "png_reader.h"
#include "image_io.h"
#include "dp.h"

namespace py = pybind11;
using namespace ngp;

static Json json_from_py(const py::handle& o) {
	if (o.is_none()) return Json();
	if (py::isinstance<py::bool_>(o)) return Json(o.cast<bool>());
	if (py::isinstance<py::int_>(o)) return Json(o.cast<long long>());
	if (py::isinstance<py::float_>(o)) return Json(o.cast<double>());
	if (py::isinstance<py::str>(o)) return Json(o.cast<std::string>());
	if (py::isinstance<py::bytes>(o)) { const std::string b = o.cast<std::string>(); return Json::binary(b.data(), b.size()); }
	if (py::isinstance<py::dict>(o)) {
		Json j = Json::object();
		for (auto kv : py::reinterpret_borrow<py::dict>(o)) j[kv.first.cast<std::string>()] = json_from_py(kv.second);
		return j;
	}
	if (py::isinstance<py::list>(o) || py::isinstance<py::tuple>(o)) {
		Json j = Json::array();
		for (auto v : o) j.push_back(json_from_py(v));
		return j;
	}
	throw std::runtime_error{"unsupported type in network config json"};
}

static py::object json_to_py(const Json& j) {
	switch (j.type()) {
		case Json::Null: return py::none();
		case Json::Bool: return py::bool_(j.boolean());
		case Json::Number: if (j.is_integer()) return py::int_((long long)j.number()); return py::float_(j.number());
		case Json::String: return py::str(j.str());
		case Json::Binary: return py::bytes((const char*)j.bin().data(), j.bin().size());
		case Json::Array: { py::list l; for (const Json& e : j.elements()) l.append(json_to_py(e)); return l; }
		default: { py::dict d; for (const auto& kv : j.items()) d[py::str(kv.first)] = json_to_py(kv.second); return d; }
	}
}

static Mat34 mat34_from_py(const py::array_t<float, py::array::c_style | py::array::forcecast>& a) {
	auto b = a.request();
	if (b.ndim != 2 || b.shape[0] < 3 || b.shape[1] != 4) throw std::runtime_error{"expected a 3x4 (or 4x4) matrix"};
	const float* p = (const float*)b.ptr;
	Mat34 m;
	for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) m.m[c * 3 + r] = p[r * 4 + c];
	return m;
}
static py::array_t<float> mat34_to_py(const Mat34& m) {
	py::array_t<float> a({3, 4});
	auto r = a.mutable_unchecked<2>();
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) r(i, j) = m.m[j * 3 + i];
	return a;
}

static Vec3 vec3_from_py(const py::object& o) {
	const std::vector<float> v = o.cast<std::vector<float>>();
	if (v.size() != 3) throw std::runtime_error{"expected a 3-vector"};
	return Vec3{v[0], v[1], v[2]};
}
static py::array_t<float> vec3_to_py(const Vec3& v) { py::array_t<float> a(3); a.mutable_data()[0] = v.x; a.mutable_data()[1] = v.y; a.mutable_data()[2] = v.z; return a; }
static Mat4 mat4_from_py(const py::array_t<float, py::array::c_style | py::array::forcecast>& a) {
	auto b = a.request();
	if (b.ndim != 2 || b.shape[0] != 4 || b.shape[1] != 4) throw std::runtime_error{"expected a 4x4 matrix"};
	const float* p = (const float*)b.ptr;
	Mat4 m;
	for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m.m[c * 4 + r] = p[r * 4 + c];
	return m;
}

// python-side value types of python_api.cu:733-802 over this build's PODs
enum class EGroundTruthRenderMode : int { Shade, Depth };
enum class ERandomMode : int { Random, Halton, Sobol, Stratified };
struct PyLens {
	ELensMode mode = ELensMode::Perspective; float params[7] = {0, 0, 0, 0, 0, 0, 0};
	static PyLens from(const NgpImageMeta& m) { PyLens l; l.mode = (ELensMode)m.lens_mode; memcpy(l.params, m.lens_params, sizeof(l.params)); return l; }
	void to(NgpImageMeta& m) const { m.lens_mode = (int)mode; memcpy(m.lens_params, params, sizeof(params)); }
};
struct PyImageMetadata { NgpImageMeta m; Vec3 light_dir; };
struct PyNerfDataset { NerfDataset* d; };

PYBIND11_MODULE(pyngp, m) {
	m.doc() = "MI355X-native Instant-NGP NeRF engine behind the blender-ngp `pyngp` API";

	py::enum_<ETestbedMode>(m, "TestbedMode").value("Nerf", ETestbedMode::Nerf).value("Sdf", ETestbedMode::Sdf).value("Image", ETestbedMode::Image).value("Volume", ETestbedMode::Volume).export_values();
	py::enum_<ERenderMode>(m, "RenderMode").value("AO", ERenderMode::AO).value("Shade", ERenderMode::Shade).value("Normals", ERenderMode::Normals).value("Positions", ERenderMode::Positions)
		.value("Depth", ERenderMode::Depth).value("Distortion", ERenderMode::Distortion).value("Cost", ERenderMode::Cost).value("Slice", ERenderMode::Slice).export_values();
	py::enum_<EGroundTruthRenderMode>(m, "GroundTruthRenderMode").value("Shade", EGroundTruthRenderMode::Shade).value("Depth", EGroundTruthRenderMode::Depth).export_values();
	py::enum_<ERandomMode>(m, "RandomMode").value("Random", ERandomMode::Random).value("Halton", ERandomMode::Halton).value("Sobol", ERandomMode::Sobol).value("Stratified", ERandomMode::Stratified).export_values();
	py::enum_<ELensMode>(m, "LensMode").value("Perspective", ELensMode::Perspective).value("OpenCV", ELensMode::OpenCV).value("FTheta", ELensMode::FTheta).value("LatLong", ELensMode::LatLong).export_values();
	py::enum_<ELossType>(m, "LossType").value("L2", ELossType::L2).value("L1", ELossType::L1).value("Mape", ELossType::Mape).value("Smape", ELossType::Smape)
		.value("Huber", ELossType::Huber).value("SmoothL1", ELossType::Huber) /* legacy name (python_api.cu:347-349) */.value("LogL1", ELossType::LogL1).value("RelativeL2", ELossType::RelativeL2).export_values();
	py::enum_<ENerfActivation>(m, "NerfActivation").value("None", ENerfActivation::None).value("ReLU", ENerfActivation::ReLU).value("Logistic", ENerfActivation::Logistic)
		.value("Exponential", ENerfActivation::Exponential).export_values();
	py::enum_<EColorSpace>(m, "ColorSpace").value("Linear", EColorSpace::Linear).value("SRGB", EColorSpace::SRGB).export_values();
	py::enum_<ETonemapCurve>(m, "TonemapCurve").value("Identity", ETonemapCurve::Identity).value("ACES", ETonemapCurve::ACES).value("Hable", ETonemapCurve::Hable)
		.value("Reinhard", ETonemapCurve::Reinhard).export_values();

	// MessagePack codec of the snapshot files (json::to_msgpack / from_msgpack in the reference): exposed for tests and tooling
	m.def("json_to_msgpack", [](const py::object& o) { const std::string b = json_from_py(o).to_msgpack(); return py::bytes(b); });
	m.def("msgpack_to_json", [](const py::bytes& b) { const std::string s = b; return json_to_py(Json::from_msgpack(s.data(), s.size())); });
	m.def("float_to_half_bits", [](const py::array_t<float, py::array::c_style | py::array::forcecast>& a) {
		py::array_t<uint16_t> r(a.size());
		for (py::ssize_t i = 0; i < a.size(); ++i) r.mutable_data()[i] = float_to_half_bits(a.data()[i]);
		return r;
	});
	m.def("half_bits_to_float", [](const py::array_t<uint16_t, py::array::c_style | py::array::forcecast>& a) {
		py::array_t<float> r(a.size());
		for (py::ssize_t i = 0; i < a.size(); ++i) r.mutable_data()[i] = half_bits_to_float(a.data()[i]);
		return r;
	});
	// host stage of the transforms.json loader (no GPU needed): exposed for tests and tooling
	m.def("decode_png", [](const std::string& path) {
		int w = 0, h = 0; std::vector<uint8_t> px;
		read_png_rgba8(path, w, h, px);
		py::array_t<uint8_t> a({h, w, 4});
		memcpy(a.mutable_data(), px.data(), px.size());
		return a;
	});
	m.def("decode_image", [](const std::string& path) {   // stbi_load(path, .., 4): PNG or JPEG by signature -> (H, W, 4) uint8
		int w = 0, h = 0; std::vector<uint8_t> px;
		read_image_rgba8(path, w, h, px);
		py::array_t<uint8_t> a({h, w, 4});
		memcpy(a.mutable_data(), px.data(), px.size());
		return a;
	});
	// test hooks for cam_adam.h (the reference's adam_optimizer.h is header-only host code without a Python face): run a sequence of steps, return every iterate
	m.def("_vec3_adam_steps", [](const py::array_t<float, py::array::c_style | py::array::forcecast>& grads, const py::array_t<float, py::array::c_style | py::array::forcecast>& lrs) {
		auto g = grads.request();
		if (g.ndim != 2 || g.shape[1] != 3 || lrs.size() != g.shape[0]) throw std::runtime_error{"grads must be (n, 3), lrs (n,)"};
		Vec3Adam a;
		py::array_t<float> out({g.shape[0], (py::ssize_t)3});
		for (py::ssize_t i = 0; i < g.shape[0]; ++i) { a.h.learning_rate = lrs.data()[i]; a.step((const float*)g.ptr + i * 3); for (int c = 0; c < 3; ++c) out.mutable_at(i, c) = a.variable[c]; }
		return out;
	});
	m.def("_rotation_adam_steps", [](const py::array_t<float, py::array::c_style | py::array::forcecast>& grads, const py::array_t<float, py::array::c_style | py::array::forcecast>& lrs) {
		auto g = grads.request();
		if (g.ndim != 2 || g.shape[1] != 3 || lrs.size() != g.shape[0]) throw std::runtime_error{"grads must be (n, 3), lrs (n,)"};
		RotationAdam a;
		py::array_t<float> out({g.shape[0], (py::ssize_t)3});
		for (py::ssize_t i = 0; i < g.shape[0]; ++i) { a.h.learning_rate = lrs.data()[i]; a.step((const float*)g.ptr + i * 3); for (int c = 0; c < 3; ++c) out.mutable_at(i, c) = a.variable[c]; }
		return out;
	});
	m.def("_angle_axis_round_trip", [](const py::array_t<float, py::array::c_style | py::array::forcecast>& angle_axis) {   // -> (3x3 matrix, angle-axis recovered from it)
		if (angle_axis.size() != 3) throw std::runtime_error{"angle_axis must have 3 elements"};
		const float* v = angle_axis.data();
		const float angle = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
		const float axis[3] = {angle > 0 ? v[0] / angle : 0.f, angle > 0 ? v[1] / angle : 0.f, angle > 0 ? v[2] / angle : 1.f};
		float mcm[9], a2, ax2[3];
		angle_axis_to_matrix(angle, axis, mcm);
		matrix_to_angle_axis(mcm, a2, ax2);
		py::array_t<float> mat({3, 3}), back(3);
		for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) mat.mutable_at(r, c) = mcm[3 * c + r];
		for (int c = 0; c < 3; ++c) back.mutable_at(c) = ax2[c] * a2;
		return py::make_tuple(mat, back);
	});
	m.def("decode_png_gray16", [](const std::string& path) {   // stbi_load_16(path, .., 1) for PNG files: the loader's depth images -> (H, W) uint16
		int w = 0, h = 0; std::vector<uint16_t> px;
		read_png_gray16(path, w, h, px);
		py::array_t<uint16_t> a({h, w});
		memcpy(a.mutable_data(), px.data(), px.size() * 2);
		return a;
	});
	m.def("decode_exr", [](const std::string& path) {   // (H, W, 4) float32, what tinyexr's LoadEXR hands the reference (tinyexr_wrapper.cu:62-112)
		int w = 0, h = 0; std::vector<float> px;
		read_exr_rgba_f32(path, w, h, px);
		py::array_t<float> a({h, w, 4});
		memcpy(a.mutable_data(), px.data(), px.size() * sizeof(float));
		return a;
	});
	m.def("load_nerf_host", [](const std::string& data_path) {
		const LoadedNerfData d = load_nerf_host(resolve_nerf_json_paths(data_path));
		py::dict out;
		out["n_images"] = d.n_images; out["paths"] = d.paths; out["scale"] = d.scale; out["aabb_scale"] = d.aabb_scale; out["from_mitsuba"] = d.from_mitsuba;
		out["offset"] = std::vector<float>{d.offset.x, d.offset.y, d.offset.z};
		out["up"] = std::vector<float>{d.up.x, d.up.y, d.up.z};
		out["render_aabb"] = std::vector<float>{d.render_aabb.min[0], d.render_aabb.min[1], d.render_aabb.min[2], d.render_aabb.max[0], d.render_aabb.max[1], d.render_aabb.max[2]};
		py::list xf, meta, px, depth, rays;
		for (size_t i = 0; i < d.n_images; ++i) {
			Mat34 s, e; memcpy(s.m, d.xforms[i].start, sizeof(s.m)); memcpy(e.m, d.xforms[i].end, sizeof(e.m));
			xf.append(py::make_tuple(mat34_to_py(s), mat34_to_py(e)));
			const NgpImageMeta& m = d.metadata[i];
			py::dict mj;
			mj["resolution"] = std::vector<int>{m.res[0], m.res[1]};
			mj["focal_length"] = std::vector<float>{m.focal_length[0], m.focal_length[1]};
			mj["principal_point"] = std::vector<float>{m.principal_point[0], m.principal_point[1]};
			mj["rolling_shutter"] = std::vector<float>(m.rolling_shutter, m.rolling_shutter + 4);
			mj["lens_mode"] = m.lens_mode; mj["lens_params"] = std::vector<float>(m.lens_params, m.lens_params + 7);
			mj["image_data_type"] = d.image_type[i];
			meta.append(mj);
			if (d.image_type[i] == 2) {   // EXR frames: RGBA fp16
				py::array_t<uint16_t> a({m.res[1], m.res[0], 4});
				memcpy(a.mutable_data(), d.pixels[i].data(), d.pixels[i].size());
				px.append(a.attr("view")("float16"));
			} else {
				py::array_t<uint8_t> a({m.res[1], m.res[0], 4});
				memcpy(a.mutable_data(), d.pixels[i].data(), d.pixels[i].size());
				px.append(a);
			}
			if (d.depth16[i].empty()) depth.append(py::none());
			else { py::array_t<uint16_t> a({m.res[1], m.res[0]}); memcpy(a.mutable_data(), d.depth16[i].data(), d.depth16[i].size() * 2); depth.append(a); }
			if (d.rays[i].empty()) rays.append(py::none());
			else { py::array_t<float> a({(py::ssize_t)m.res[1], (py::ssize_t)m.res[0], (py::ssize_t)6}); memcpy(a.mutable_data(), d.rays[i].data(), d.rays[i].size() * sizeof(NgpRay)); rays.append(a); }
		}
		out["xforms"] = xf; out["metadata"] = meta; out["pixels"] = px; out["depth16"] = depth; out["rays"] = rays;
		out["depth_scale"] = d.depth_scale; out["has_rays"] = d.has_rays; out["is_hdr"] = d.is_hdr;
		out["envmap_resolution"] = std::vector<int>{d.envmap_resolution[0], d.envmap_resolution[1]};
		if (d.envmap_data.empty()) out["envmap"] = py::none();
		else { py::array_t<float> e({(size_t)d.envmap_resolution[1], (size_t)d.envmap_resolution[0], (size_t)4}); memcpy(e.mutable_data(), d.envmap_data.data(), d.envmap_data.size() * 4); out["envmap"] = e; }
		return out;
	});
	// data-parallel plumbing that needs no GPU (tests): the shared-memory counter exchange of the ranks of one node, and the RCCL binding probe
	py::class_<ShmCounterExchange>(m, "ShmCounterExchange")
		.def(py::init<uint32_t, uint32_t, const std::string&, double>(), py::call_guard<py::gil_scoped_release>(), py::arg("rank"), py::arg("world_size"), py::arg("key"), py::arg("timeout_s") = 60.0)
		.def("all_sum", [](ShmCounterExchange& x, uint64_t step, double a, double b, double c) { const double in[3] = {a, b, c}; double out[3]; { py::gil_scoped_release rel; x.all_sum(step, in, out); } return py::make_tuple(out[0], out[1], out[2]); })
		.def("publish_blob", [](ShmCounterExchange& x, const py::bytes& b) { const std::string s = b; if (s.size() != 128) throw std::runtime_error{"128 bytes"}; x.publish_blob((const uint8_t*)s.data()); })
		.def("fetch_blob", [](ShmCounterExchange& x) { uint8_t b[128]; { py::gil_scoped_release rel; x.fetch_blob(b); } return py::bytes((const char*)b, 128); })
		.def("barrier", &ShmCounterExchange::barrier, py::call_guard<py::gil_scoped_release>());
	m.def("rccl_available", []() { return ngp_rccl_available() != 0; });
	m.def("free_temporary_memory", []() {});  // python_api.cu:309 (arenas are RAII buffers here)
	m.def("device_memory_allocated", []() { return DeviceBuffer::total_allocated(); });

	// ---- Blender add-on request schema (python_api.cu:392-538)
	py::enum_<ECameraModel>(m, "CameraModel").value("Perspective", ECameraModel::Perspective).value("SphericalQuadrilateral", ECameraModel::SphericalQuadrilateral)
		.value("QuadrilateralHexahedron", ECameraModel::QuadrilateralHexahedron).export_values();
	py::enum_<EMaskMode>(m, "MaskMode").value("Add", EMaskMode::Add).value("Subtract", EMaskMode::Subtract).export_values();
	py::enum_<EMaskShape>(m, "MaskShape").value("Box", EMaskShape::Box).value("Cylinder", EMaskShape::Cylinder).value("Sphere", EMaskShape::Sphere).export_values();

	py::class_<BoundingBox>(m, "BoundingBox")
		.def(py::init<>())
		.def(py::init([](const py::object& a, const py::object& b) { return BoundingBox(vec3_from_py(a), vec3_from_py(b)); }))
		.def("center", [](const BoundingBox& b) { return vec3_to_py(b.center()); })
		.def("contains", [](const BoundingBox& b, const py::object& p) { return b.contains(vec3_from_py(p)); })
		.def("diag", [](const BoundingBox& b) { return vec3_to_py(b.diag()); })
		.def("enlarge", [](BoundingBox& b, const py::object& p) { if (py::isinstance<BoundingBox>(p)) b.enlarge(p.cast<BoundingBox>()); else b.enlarge(vec3_from_py(p)); })
		.def("inflate", &BoundingBox::inflate)
		.def("distance", [](const BoundingBox& b, const py::object& p) { return std::sqrt(b.distance_sq(vec3_from_py(p))); })
		.def("distance_sq", [](const BoundingBox& b, const py::object& p) { return b.distance_sq(vec3_from_py(p)); })
		.def("signed_distance", [](const BoundingBox& b, const py::object& p) { return b.signed_distance(vec3_from_py(p)); })
		.def("intersection", &BoundingBox::intersection)
		.def("intersects", &BoundingBox::intersects)
		.def("ray_intersect", [](const BoundingBox& b, const py::object& pos, const py::object& dir) { float t[2]; b.ray_intersect(vec3_from_py(pos), vec3_from_py(dir), t); py::array_t<float> a(2); a.mutable_data()[0] = t[0]; a.mutable_data()[1] = t[1]; return a; })
		.def("get_vertices", [](const BoundingBox& b) { py::list l; for (int i = 0; i < 8; ++i) l.append(vec3_to_py(Vec3{(i & 4) ? b.max.x : b.min.x, (i & 2) ? b.max.y : b.min.y, (i & 1) ? b.max.z : b.min.z})); return l; })
		.def("relative_pos", [](const BoundingBox& b, const py::object& p) { return vec3_to_py(b.relative_pos(vec3_from_py(p))); })
		.def_property("min", [](const BoundingBox& b) { return vec3_to_py(b.min); }, [](BoundingBox& b, const py::object& v) { b.min = vec3_from_py(v); })
		.def_property("max", [](const BoundingBox& b) { return vec3_to_py(b.max); }, [](BoundingBox& b, const py::object& v) { b.max = vec3_from_py(v); });

	py::class_<Quadrilateral3D>(m, "Quadrilateral3D")
		.def_static("Zero", []() { return Quadrilateral3D{}; })
		.def(py::init([](const py::object& tl, const py::object& tr, const py::object& bl, const py::object& br) { return Quadrilateral3D{vec3_from_py(tl), vec3_from_py(tr), vec3_from_py(bl), vec3_from_py(br)}; }),
			py::arg("tl"), py::arg("tr"), py::arg("bl"), py::arg("br"))
		.def("center", [](const Quadrilateral3D& q) { return vec3_to_py(q.center()); });
	py::class_<QuadrilateralHexahedron>(m, "QuadrilateralHexahedronConfig")
		.def_static("Zero", []() { return QuadrilateralHexahedron{}; })
		.def(py::init([](const Quadrilateral3D& f, const Quadrilateral3D& b) { return QuadrilateralHexahedron{f, b}; }), py::arg("front"), py::arg("back"))
		.def("center", [](const QuadrilateralHexahedron& q) { return vec3_to_py(q.center()); });
	py::class_<SphericalQuadrilateral>(m, "SphericalQuadrilateralConfig")
		.def_static("Zero", []() { return SphericalQuadrilateral{}; })
		.def(py::init([](float w, float h, float c) { return SphericalQuadrilateral{w, h, c}; }), py::arg("width"), py::arg("height"), py::arg("curvature"))
		.def_readwrite("width", &SphericalQuadrilateral::width).def_readwrite("height", &SphericalQuadrilateral::height).def_readwrite("curvature", &SphericalQuadrilateral::curvature);

	py::class_<Mask3D>(m, "Mask3D")
		.def_static("Box", [](const py::object& dims, const py::array_t<float, py::array::c_style | py::array::forcecast>& t, EMaskMode mode, float feather, float opacity) {
				return Mask3D::Box(vec3_from_py(dims), mat4_from_py(t), mode, feather, opacity); }, py::arg("dims"), py::arg("transform"), py::arg("mode"), py::arg("feather"), py::arg("opacity"))
		.def_static("Cylinder", [](float radius, float height, const py::array_t<float, py::array::c_style | py::array::forcecast>& t, EMaskMode mode, float feather, float opacity) {
				return Mask3D::Cylinder(radius, height, mat4_from_py(t), mode, feather, opacity); }, py::arg("radius"), py::arg("height"), py::arg("transform"), py::arg("mode"), py::arg("feather"), py::arg("opacity"))
		.def_static("Sphere", [](float radius, const py::array_t<float, py::array::c_style | py::array::forcecast>& t, EMaskMode mode, float feather, float opacity) {
				return Mask3D::Sphere(radius, mat4_from_py(t), mode, feather, opacity); }, py::arg("radius"), py::arg("transform"), py::arg("mode"), py::arg("feather"), py::arg("opacity"));

	py::class_<DownsampleInfo>(m, "DownsampleInfo")
		.def_static("MakeFromMip", [](const std::vector<int>& res, uint32_t mip) { if (res.size() != 2) throw std::runtime_error{"resolution must have 2 entries"}; return DownsampleInfo::MakeFromMip(res[0], res[1], mip); },
			py::arg("resolution"), py::arg("mip"));

	py::class_<RenderOutputProperties>(m, "RenderOutputProperties")
		.def(py::init([](const std::vector<int>& res, const DownsampleInfo& ds, uint32_t spp, EColorSpace cs, ETonemapCurve tc, float exposure, const std::vector<float>& bg, bool flip_y) {
				if (res.size() != 2 || bg.size() != 4) throw std::runtime_error{"resolution needs 2 entries, background_color 4"};
				RenderOutputProperties o;
				o.resolution[0] = res[0]; o.resolution[1] = res[1]; o.ds = ds; o.spp = spp; o.color_space = cs; o.tonemap_curve = tc; o.exposure = exposure;
				for (int i = 0; i < 4; ++i) o.background_color[i] = bg[i];
				o.flip_y = flip_y;
				return o;
			}), py::arg("resolution"), py::arg("ds"), py::arg("spp"), py::arg("color_space"), py::arg("tonemap_curve"), py::arg("exposure"), py::arg("background_color"), py::arg("flip_y"));

	py::class_<RenderModifiersDescriptor>(m, "RenderModifiers")
		.def(py::init([](const std::vector<Mask3D>& masks) { return RenderModifiersDescriptor{masks}; }), py::arg("masks"));

	py::class_<RenderCameraProperties>(m, "RenderCameraProperties")
		.def(py::init([](const py::array_t<float, py::array::c_style | py::array::forcecast>& transform, ECameraModel model, float focal_length, float near_distance, float aperture_size, float focus_z,
		                 const SphericalQuadrilateral& sq, const QuadrilateralHexahedron& qh) {
				RenderCameraProperties c;
				c.transform = mat34_from_py(transform); c.model = model; c.focal_length = focal_length; c.near_distance = near_distance; c.aperture_size = aperture_size; c.focus_z = focus_z;
				c.spherical_quadrilateral = sq; c.quadrilateral_hexahedron = qh;
				return c;
			}), py::arg("transform"), py::arg("model"), py::arg("focal_length"), py::arg("near_distance"), py::arg("aperture_size"), py::arg("focus_z"), py::arg("spherical_quadrilateral"),
			py::arg("quadrilateral_hexahedron"))
		.def("__eq__", [](const RenderCameraProperties& a, const RenderCameraProperties& b) { return a == b; })
		.def("__ne__", [](const RenderCameraProperties& a, const RenderCameraProperties& b) { return a != b; });

	py::class_<NerfDescriptor>(m, "NerfDescriptor")
		.def(py::init([](const std::string& path, const BoundingBox& aabb, const py::array_t<float, py::array::c_style | py::array::forcecast>& transform, const RenderModifiersDescriptor& mods, float opacity) {
				NerfDescriptor d;
				d.snapshot_path = path; d.aabb = aabb; d.transform = mat4_from_py(transform); d.modifiers = mods; d.opacity = opacity;
				return d;
			}), py::arg("snapshot_path_str"), py::arg("aabb"), py::arg("transform"), py::arg("modifiers"), py::arg("opacity"));

	py::class_<RenderRequest>(m, "RenderRequest")
		.def(py::init([](const RenderOutputProperties& output, const RenderCameraProperties& camera, const RenderModifiersDescriptor& modifiers, const std::vector<NerfDescriptor>& nerfs, const BoundingBox& aabb) {
				return RenderRequest{output, camera, modifiers, nerfs, aabb};
			}), py::arg("output"), py::arg("camera"), py::arg("modifiers"), py::arg("nerfs"), py::arg("aabb"));

	// python_api.cu:733-742, 771-802: value views.  Like the reference's (std::vector members cross pybind11 by COPY), what Python gets are copies.
	py::class_<PyLens>(m, "Lens")
		.def(py::init<>())
		.def_readwrite("mode", &PyLens::mode)
		.def_property_readonly("params", [](py::object& obj) { PyLens& o = obj.cast<PyLens&>(); return py::array(py::dtype::of<float>(), {7}, {sizeof(float)}, o.params, obj); });
	py::class_<PyImageMetadata>(m, "TrainingImageMetadata")
		.def_property("focal_length", [](PyImageMetadata& t) { py::array_t<float> a(2); a.mutable_data()[0] = t.m.focal_length[0]; a.mutable_data()[1] = t.m.focal_length[1]; return a; },
			[](PyImageMetadata& t, const std::vector<float>& v) { t.m.focal_length[0] = v.at(0); t.m.focal_length[1] = v.at(1); })
		.def_property("principal_point", [](PyImageMetadata& t) { py::array_t<float> a(2); a.mutable_data()[0] = t.m.principal_point[0]; a.mutable_data()[1] = t.m.principal_point[1]; return a; },
			[](PyImageMetadata& t, const std::vector<float>& v) { t.m.principal_point[0] = v.at(0); t.m.principal_point[1] = v.at(1); })
		.def_property("rolling_shutter", [](PyImageMetadata& t) { py::array_t<float> a(4); memcpy(a.mutable_data(), t.m.rolling_shutter, 16); return a; },
			[](PyImageMetadata& t, const std::vector<float>& v) { for (int k = 0; k < 4; ++k) t.m.rolling_shutter[k] = v.at(k); })
		.def_property("light_dir", [](PyImageMetadata& t) { return vec3_to_py(t.light_dir); }, [](PyImageMetadata& t, const py::object& v) { t.light_dir = vec3_from_py(v); })
		.def_property("lens", [](PyImageMetadata& t) { return PyLens::from(t.m); }, [](PyImageMetadata& t, const PyLens& l) { l.to(t.m); })
		.def_property("camera_distortion", [](PyImageMetadata& t) { return PyLens::from(t.m); }, [](PyImageMetadata& t, const PyLens& l) { l.to(t.m); })   // legacy name
		.def_property_readonly("resolution", [](PyImageMetadata& t) { return std::vector<int>{t.m.res[0], t.m.res[1]}; });
	py::class_<PyNerfDataset>(m, "NerfDataset")
		.def_property_readonly("metadata", [](PyNerfDataset& d) { std::vector<PyImageMetadata> v; for (const NgpImageMeta& m : d.d->metadata) v.push_back(PyImageMetadata{m, Vec3{}}); return v; })
		.def_property_readonly("transforms", [](PyNerfDataset& d) { py::list l; for (const NgpXForm& x : d.d->xforms) { Mat34 a, b; memcpy(a.m, x.start, 48); memcpy(b.m, x.end, 48); l.append(py::make_tuple(mat34_to_py(a), mat34_to_py(b))); } return l; })
		.def_property_readonly("paths", [](PyNerfDataset& d) { return d.d->paths; })
		.def_property_readonly("render_aabb", [](PyNerfDataset& d) { const NgpAabb& a = d.d->render_aabb; return BoundingBox(Vec3{a.min[0], a.min[1], a.min[2]}, Vec3{a.max[0], a.max[1], a.max[2]}); })
		.def_property_readonly("render_aabb_to_local", [](PyNerfDataset&) { py::array_t<float> a({3, 3}); for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) a.mutable_at(r, c) = r == c ? 1.f : 0.f; return a; })   // the loader of this build reads no `render_aabb_to_local`-style key (nerf_loader.cu sets identity too)
		.def_property_readonly("up", [](PyNerfDataset& d) { return vec3_to_py(d.d->up); })
		.def_property_readonly("offset", [](PyNerfDataset& d) { return vec3_to_py(d.d->offset); })
		.def_property_readonly("n_images", [](PyNerfDataset& d) { return d.d->n_images; })
		.def_property_readonly("envmap_resolution", [](PyNerfDataset& d) { return std::vector<int>{d.d->envmap_resolution[0], d.d->envmap_resolution[1]}; })   // python_api.cu:797
		.def_property_readonly("scale", [](PyNerfDataset& d) { return d.d->scale; })
		.def_property_readonly("aabb_scale", [](PyNerfDataset& d) { return d.d->aabb_scale; })
		// nerf_loader.h:94-99 (read from transforms.json by the loader; writable here for datasets built with create_empty_nerf_dataset — takes effect at the next reload_network / reset_network)
		.def_property("n_extra_learnable_dims", [](PyNerfDataset& d) { return d.d->n_extra_learnable_dims; }, [](PyNerfDataset& d, uint32_t n) { d.d->n_extra_learnable_dims = n; })
		.def_property_readonly("has_light_dirs", [](PyNerfDataset& d) { return d.d->has_light_dirs; })
		.def_property_readonly("n_extra_dims", [](PyNerfDataset& d) { return d.d->n_extra_dims(); })
		.def("set_light_dirs", [](PyNerfDataset& d, const py::array_t<float, py::array::c_style | py::array::forcecast>& dirs) {   // per image, NGP frame (what `driver_parameters` gives the loader)
				if (dirs.ndim() != 2 || dirs.shape(1) != 3 || (size_t)dirs.shape(0) != d.d->n_images) throw std::runtime_error{"set_light_dirs: an (n_images, 3) array"};
				d.d->light_dirs.resize(d.d->n_images);
				for (size_t i = 0; i < d.d->n_images; ++i) d.d->light_dirs[i] = Vec3{dirs.at(i, 0), dirs.at(i, 1), dirs.at(i, 2)};
				d.d->has_light_dirs = true; d.d->n_extra_learnable_dims = 0;
			}, py::arg("light_dirs"))
		.def_property_readonly("from_mitsuba", [](PyNerfDataset& d) { return d.d->from_mitsuba; })
		.def_property_readonly("is_hdr", [](PyNerfDataset& d) { return d.d->is_hdr; });

	// a Testbed that is collected while an async render is in flight waits for the worker, and the worker needs the GIL for its callback
	struct TestbedDeleter { void operator()(Testbed* t) const { { py::gil_scoped_release rel; t->bl_wait_for_renders(); } delete t; } };
	py::class_<Testbed, std::unique_ptr<Testbed, TestbedDeleter>> testbed(m, "Testbed");
	testbed
		.def(py::init<ETestbedMode>(), py::arg("mode") = ETestbedMode::Nerf)
		.def(py::init([](ETestbedMode mode, const std::string& data_path, const py::object& network_config) {   // python_api.cu:541-543 (testbed.cu:3108-3122)
				std::unique_ptr<Testbed, TestbedDeleter> t(new Testbed(mode));
				t->load_training_data(data_path);
				if (py::isinstance<py::str>(network_config)) t->reload_network_from_file(network_config.cast<std::string>());
				else t->reload_network_from_json(json_from_py(network_config));
				return t;
			}), py::arg("mode"), py::arg("data_path"), py::arg("network_config"))
		.def("load_training_data", &Testbed::load_training_data, py::call_guard<py::gil_scoped_release>(), py::arg("path"))
		.def("create_empty_nerf_dataset", &Testbed::create_empty_nerf_dataset, py::arg("n_images"), py::arg("aabb_scale") = 1, py::arg("is_hdr") = false)
		.def("frame", &Testbed::frame, py::call_guard<py::gil_scoped_release>())
		.def("train", &Testbed::train, py::call_guard<py::gil_scoped_release>(), py::arg("batch_size"))
		.def("reset", &Testbed::reset, py::arg("reset_density_grid") = true)
		.def("reload_network_from_file", &Testbed::reload_network_from_file, py::arg("path") = "")
		.def("reload_network_from_json", [](Testbed& t, const py::object& json, const std::string& base) { t.reload_network_from_json(json_from_py(json), base); },
			py::arg("json"), py::arg("config_base_path") = "")
		.def("save_snapshot", &Testbed::save_snapshot, py::arg("path"), py::arg("include_optimizer_state") = false)
		.def("load_snapshot", &Testbed::load_snapshot, py::arg("path"))
		// entry points of scripts/run.py that lie outside the hot path (SURVEY §8 out of scope): present so that a driver fails loudly, not with AttributeError
		.def("render_with_rolling_shutter", [](Testbed& t, const py::array_t<float, py::array::c_style | py::array::forcecast>& m0, const py::array_t<float, py::array::c_style | py::array::forcecast>& m1,
		                                       const std::vector<float>& rolling_shutter, int width, int height, int spp, bool linear) {   // python_api.cu:262-275, 584-593
				if (rolling_shutter.size() != 4) throw std::runtime_error{"rolling_shutter takes 4 floats [A, B, C, D]"};
				const Mat34 start = mat34_from_py(m0), end = mat34_from_py(m1);   // buffer requests touch refcounts: before the GIL is released
				py::array_t<float> result({height, width, 4});   // the frame is written straight into the array that is returned (no intermediate host copy)
				float* dst = result.mutable_data();
				{ py::gil_scoped_release rel; t.render_with_rolling_shutter_to_cpu(start, end, rolling_shutter.data(), width, height, spp, linear, dst); }
				return result;
			}, "Renders an image at the requested resolution. Does not require a window. Supports rolling shutter, with per ray time being computed as A+B*u+C*v+D*t for [A,B,C,D]",
			py::arg("transform_matrix_start"), py::arg("transform_matrix_end"), py::arg("rolling_shutter") = std::vector<float>{0.f, 0.f, 0.f, 0.f}, py::arg("width") = 1920, py::arg("height") = 1080,
			py::arg("spp") = 1, py::arg("linear") = true)
		.def("destroy_window", [](Testbed&) {})                                                                          // python_api.cu:594 (there never is one)
		.def("calculate_iou", [](Testbed&, uint64_t, float, bool, bool) -> float { throw std::runtime_error{"calculate_iou: the SDF ground-truth metric (mesh BVH, testbed_sdf.cu) is not part of this build"}; },
			py::arg("n_samples") = 128 * 1024 * 1024, py::arg("scale_existing_results_factor") = 0.0f, py::arg("blocking") = true, py::arg("force_use_octree") = true)
		.def("compute_and_save_png_slices", [](Testbed&, const std::string&, py::object, py::object, float, float, bool) { throw std::runtime_error{"compute_and_save_png_slices: density-slice export is not part of this build"}; },
			py::arg("filename"), py::arg("resolution") = py::none(), py::arg("aabb") = py::none(), py::arg("thresh") = 3.4e38f, py::arg("density_range") = 4.f, py::arg("flip_y_and_z_axes") = false)
		.def("compute_marching_cubes_mesh", [](Testbed&, py::object, py::object, float) -> py::dict { throw std::runtime_error{"compute_marching_cubes_mesh: mesh extraction is not part of this build"}; },
			py::arg("resolution") = py::none(), py::arg("aabb") = py::none(), py::arg("thresh") = 3.4e38f)
		.def("first_training_view", &Testbed::first_training_view).def("last_training_view", &Testbed::last_training_view)
		.def("previous_training_view", &Testbed::previous_training_view).def("next_training_view", &Testbed::next_training_view)
		.def("crop_box", [](Testbed& t, bool nerf_space) { return mat34_to_py(t.crop_box(nerf_space)); }, py::arg("nerf_space") = true)             // python_api.cu:729-731
		.def("set_crop_box", [](Testbed& t, const py::array_t<float, py::array::c_style | py::array::forcecast>& m, bool nerf_space) { t.set_crop_box(mat34_from_py(m), nerf_space); }, py::arg("matrix"), py::arg("nerf_space") = true)
		.def("crop_box_corners", [](Testbed& t, bool nerf_space) { py::list l; for (const Vec3& v : t.crop_box_corners(nerf_space)) l.append(vec3_to_py(v)); return l; }, py::arg("nerf_space") = true)
		.def("want_repl", [](Testbed&) { return false; })                                                                  // python_api.cu:363 (GUI only)
		.def("init_window", [](Testbed&, int, int, bool, bool) { throw std::runtime_error{"init_window: this build has no GUI (windowless rendering only)"}; },
			py::arg("width"), py::arg("height"), py::arg("hidden") = false, py::arg("second_window") = false)
		.def("load_camera_path", [](Testbed&, const std::string&) { throw std::runtime_error{"load_camera_path: camera paths are not part of this build (set the camera per frame with set_nerf_camera_matrix)"}; }, py::arg("path"))
		.def("compute_and_save_marching_cubes_mesh", [](Testbed&, const std::string&, py::object, py::object, float) { throw std::runtime_error{"compute_and_save_marching_cubes_mesh: mesh extraction is not part of this build"}; },
			py::arg("filename"), py::arg("resolution") = py::none(), py::arg("aabb") = py::none(), py::arg("thresh") = 2.5f)
		.def("n_params", &Testbed::n_params)
		.def("n_encoding_params", &Testbed::n_encoding_params)
		.def("render", [](Testbed& t, int width, int height, int spp, bool linear, float start_t, float end_t, float, float) {
				// python_api.cu:131-165: start_t >= 0 animates along the loaded camera path (set_camera_from_time, smoothing, per-spp shutter interpolation).  Camera
				// paths are not part of this build (load_camera_path throws): a still frame returned for a path request would be silently wrong
				if (start_t >= 0.f /* the reference's path_animation_enabled (python_api.cu:139): end_t alone is ignored there too */) throw std::runtime_error{"render(start_t >= 0): camera-path animation is not part of this build; set the camera per frame with set_nerf_camera_matrix and call render() with start_t = end_t = -1"};
				py::array_t<float> result({height, width, 4});   // the frame is written straight into the array that is returned (no intermediate host copy)
				float* dst = result.mutable_data();
				{ py::gil_scoped_release rel; t.render_to_cpu(width, height, spp, linear, dst); }
				return result;
			}, py::arg("width") = 1920, py::arg("height") = 1080, py::arg("spp") = 1, py::arg("linear") = true, py::arg("start_t") = -1.f, py::arg("end_t") = -1.f,
			py::arg("fps") = 30.f, py::arg("shutter_fraction") = 1.0f)
		// plumbing configs
		.def("set_image_data", [](Testbed& t, const py::array_t<float, py::array::c_style | py::array::forcecast>& img) {
				auto b = img.request();
				if (b.ndim != 3 || b.shape[2] != 4) throw std::runtime_error{"image should be (H,W,4) float32"};
				t.set_image_data((int)b.shape[1], (int)b.shape[0], (const float*)b.ptr);
			}, py::arg("img"), "Image mode: fit this linear RGBA float image (what load_exr_image leaves on the device)")
		.def("compute_image_mse", &Testbed::compute_image_mse, py::call_guard<py::gil_scoped_release>(), py::arg("quantize") = false)   // python_api.cu:600
		.def("override_sdf_training_data", [](Testbed& t, const py::array_t<float, py::array::c_style | py::array::forcecast>& points, const py::array_t<float, py::array::c_style | py::array::forcecast>& distances) {
				auto pb = points.request(), db = distances.request();
				if (pb.ndim != 2 || db.ndim != 1 || pb.shape[0] != db.shape[0] || pb.shape[1] != 3) throw std::runtime_error{"Invalid Points<->Distances data"};
				t.override_sdf_training_data((const float*)pb.ptr, (const float*)db.ptr, (size_t)pb.shape[0]);
			}, "Override the training data for learning a signed distance function")                                                  // python_api.cu:605
		.def("gridmlp_inference", [](Testbed& t, const py::array_t<float, py::array::c_style | py::array::forcecast>& pos) {   // tooling / tests: network outputs 0..3 at the given positions
				auto b = pos.request();
				const uint32_t n_dims = t.gridmlp_n_dims();
				if (b.ndim != 2 || (uint32_t)b.shape[1] != n_dims) throw std::runtime_error{"positions must be (N, n_dims)"};
				const uint32_t n = (uint32_t)b.shape[0];
				DeviceBuffer d_pos, d_out;
				d_pos.resize((size_t)n * n_dims * 4); d_out.resize((size_t)n * 8);
				d_pos.copy_from_host(b.ptr, (size_t)n * n_dims * 4);
				t.join_side_ema();
				if (ngp_hip_gridmlp_forward(t.stream(), n_dims, t.m_desc_gpu.as<NgpNetDesc>(), t.m_inference_params.as<uint16_t>(), d_pos.as<float>(), n_dims, n, d_out.as<uint16_t>(), 4, nullptr))
					throw std::runtime_error{ngp_hip_last_error()};
				t.sync();
				std::vector<uint16_t> h((size_t)n * 4);
				d_out.copy_to_host(h.data(), h.size() * 2);
				py::array_t<float> out({(py::ssize_t)n, (py::ssize_t)4});
				for (size_t i = 0; i < h.size(); ++i) out.mutable_data()[i] = half_bits_to_float(h[i]);
				return out;
			}, py::arg("positions"))
		.def_property_readonly("image_resolution", [](Testbed& t) { return std::vector<int>{t.m_image.resolution[0], t.m_image.resolution[1]}; })
		.def("request_nerf_render_sync", [](Testbed& t, const RenderRequest& req) {  // python_api.cu:233-260, 581
				py::array_t<float> result({req.output.resolution[1], req.output.resolution[0], 4});
				float* dst = result.mutable_data();
				bool rendered;
				{ py::gil_scoped_release rel; rendered = t.bl_request_nerf_render_sync(req, dst); }
				if (!rendered) memset(dst, 0, (size_t)result.size() * sizeof(float));   // the reference returns the untouched (zero) array while another render is in flight
				return result;
			}, "Requests a nerf render frame.", py::arg("render_request"))
		.def("request_nerf_render_async", [](Testbed& t, const RenderRequest& req, const py::function& render_callback) {  // python_api.cu:192-231, 577-580
				if (!t.bl_try_begin_render()) return;   // a render is already in flight: the request is dropped (:218-220)
				struct EndRender { Testbed& t; bool armed = true; ~EndRender() { if (armed) t.bl_end_render(); } void now() { if (armed) { armed = false; t.bl_end_render(); } } };
				auto cb = std::make_shared<py::function>(render_callback);
				try {
					if (t.m_autofocus) t.autofocus();   // :222-224
					t.bl_start_async([&t, req, cb]() mutable {
						EndRender end{t};   // whatever happens below, the busy flag is cleared
						std::vector<float> px;
						std::string error;
						try {
							RenderBuffer& rb = t.m_bl_render_surface;
							rb.resize(req.output.resolution[0], req.output.resolution[1]);
							rb.reset_accumulation();
							t.bl_render_frame(rb, req);
							px.resize((size_t)req.output.resolution[0] * req.output.resolution[1] * 4);
							t.download(rb.surface.data(), px.size() * 4, px.data());   // through the Testbed's pinned staging buffer (testbed.cpp download)
						} catch (const std::exception& e) { error = e.what(); } catch (...) { error = "unknown error"; }
						end.now();   // before the callback, so that it may queue the next request (from this very thread: nothing joins it)
						py::gil_scoped_acquire acquire;
						try {
							if (error.empty()) {
								py::array_t<float> result({req.output.resolution[1], req.output.resolution[0], 4});
								memcpy(result.mutable_data(), px.data(), px.size() * sizeof(float));
								(*cb)(result);
							} else {
								fprintf(stderr, "request_nerf_render_async failed: %s\n", error.c_str());
							}
						} catch (py::error_already_set& e) { fprintf(stderr, "render callback raised: %s\n", e.what()); }
						cb.reset();   // drop the Python reference while the GIL is held
					});
				} catch (...) { t.bl_end_render(); throw; }
			}, "Requests a nerf render frame.", py::arg("render_request"), py::arg("render_callback"))
		.def("wait_for_render", [](Testbed& t) { py::gil_scoped_release rel; t.bl_wait_for_renders(); })
		.def_readwrite("netx_scalar_kernels", &Testbed::m_netx_scalar_kernels, "network variants (extra dims, 0 / 1 / 3 hidden colour layers): True runs the scalar checker kernels (bit-compatible with the oracle's restatement) instead of the MFMA kernels")
		.def_readonly("bl_render_samples", &Testbed::m_bl_render_samples)
		.def_readonly("bl_render_passes", &Testbed::m_bl_render_passes)
		.def_readwrite("compact_backward", &Testbed::m_compact_backward, "training step (base network family): False (default) = the backward pass over all B slots as the reference does; True = the backward pass runs over the samples whose loss gradient is not zero in all four channels (ray tails in fp16: 30-45 % of a batch are), moved to the front of a second set of buffers and counted on the device — same hash-grid gradients bit for bit, MLP weight gradients the same sums in another order.  The kernels of the backward pass get 5-20 us faster each, the step does not (profiles/r05_experiments.md section 8)")
		.def_property_readonly("backward_live_fraction", &Testbed::backward_live_fraction, "live samples / batch size of the last step whose backward pass ran over the live samples (drains the stream; 1.0 if none did)")
		.def_readwrite("ema_on_side_stream", &Testbed::m_ema_on_side_stream, "optimizer step: False (default) = one launch, Adam + Ema, on the training chain; True = the Ema stage (what renderers and snapshots read) runs on the second stream beside the next step's network pass (a gain on the fox photographs only: profiles/r05_experiments.md section 7).  The same bits either way")
		.def_readwrite("bl_fused_passes", &Testbed::m_bl_fused_passes, "Blender renderer pass loop: True (default) = one fused launch (march + cull + compact + per-NeRF lists) and one host-mailbox poll per pass on the stock tracer's sample budget; False = the reference's launch sequence with its two blocking read-backs per pass (same pixels)")
		.def_readwrite("bl_reference_schedule", &Testbed::m_bl_reference_schedule, "fused pass loop on the unfused loop's schedule (n_steps from the rays that entered the pass, no resting rays): the same frame bit for bit; the fork's sampler depends on where the pass boundaries fall")
		.def_readwrite("bl_max_skips_per_pass", &Testbed::m_bl_max_skips_per_pass)
		.def_readwrite("bl_max_steps_per_pass", &Testbed::m_bl_max_steps_per_pass)
		.def_readwrite("bl_pass_samples_factor", &Testbed::m_bl_pass_samples_factor)
		.def("set_nerf_camera_matrix", [](Testbed& t, const py::array_t<float, py::array::c_style | py::array::forcecast>& cam) { t.set_nerf_camera_matrix(mat34_from_py(cam)); })
		.def("reset_camera", &Testbed::reset_camera)
		.def("reset_accumulation", [](Testbed& t, bool, bool) { t.m_windowless_render_surface.reset_accumulation(); }, py::arg("due_to_camera_movement") = false, py::arg("immediate_redraw") = true)
		// data-parallel extension
		.def("set_distributed", &Testbed::set_distributed, py::arg("rank"), py::arg("world_size"))
		.def("init_data_parallel", &Testbed::init_data_parallel, py::call_guard<py::gil_scoped_release>(), py::arg("rank"), py::arg("world_size"), py::arg("key") = std::string("0"), py::arg("strong_scaling") = false,
			"One process per GPU of one node: after this call frame() / train() run the data-parallel step (shared-memory counter exchange, RCCL gradient all-reduce over xGMI). "
			"`key` names the rendezvous and must be the same on every rank of the job (e.g. MASTER_PORT). strong_scaling: train(B) back-propagates B / world_size samples per rank, and rays_per_batch is divided by the world size here (once per call), so the first data-parallel step marches the global rays of the single-rank step.")
		.def("shutdown_data_parallel", &Testbed::shutdown_data_parallel, py::call_guard<py::gil_scoped_release>(), "tears the communicator down; NO collective (a rank may leave alone).  Call dp_gather_optimizer_state() on all ranks first if the Testbed is to train on or to save its optimizer state")
		.def("dp_gather_optimizer_state", &Testbed::dp_gather_optimizer_state, py::call_guard<py::gil_scoped_release>(), "COLLECTIVE (every rank): all-gather of the fp32 master weights and Adam moments the sharded optimizer step keeps current only inside each rank's shard; required before save_snapshot(include_optimizer_state=True) and before training on after shutdown_data_parallel")
		.def("set_render_shard", &Testbed::set_render_shard, py::arg("rank"), py::arg("world_size"),
			"render() traces only the rows of shard `rank` of `world_size` (rows [rank * ceil(H / world), ...)); the rest of the returned frame is background. With init_data_parallel the split and the gather happen inside render().")
		.def("render_shard_rows", [](Testbed& t, int height) { int a, b; t.render_shard_rows(height, a, b); return py::make_tuple(a, b); }, py::arg("height"))
		.def_property("dp_sharded_optimizer", [](Testbed& t) { return t.m_dp_sharded_optimizer; }, &Testbed::set_dp_sharded_optimizer, "data-parallel step: reduce-scatter (fp32 sums) -> Adam on this rank's 1 / world of the parameters -> all-gather of the fp16 weights (default); False: fp16 all-reduce of the gradients, the whole optimizer step on every rank.  Settable only while no communicator is live (before init_data_parallel)")
		.def_property("dp_sharded_ema", [](Testbed& t) { return t.m_dp_sharded_ema; }, &Testbed::set_dp_sharded_ema, "sharded data-parallel step: the Ema stage runs on this rank's shard only (one launch with Adam) and the fp16 inference weights are gathered on demand — render() gathers by itself when it is a collective (render_sharded), otherwise call dp_gather_inference_params() on all ranks before render() / save_snapshot().  False: Ema over all parameters on every rank (rounds 3-4).  Settable only while no communicator is live")
		.def_readwrite("dp_fp16_wire", &Testbed::m_dp_fp16_wire, "sharded data-parallel step: gradients cross the wire as fp16 slices (point to point, ngp_rccl_alltoall_f16) and are summed by their owner in rank order in fp32 with one fp16 rounding; False: widen -> fp32 reduce-scatter -> narrow (twice the bytes).  Set it alike on every rank, between two steps")
		.def("dp_gather_inference_params", &Testbed::dp_gather_inference_params, py::call_guard<py::gil_scoped_release>(), "COLLECTIVE (every rank): all-gather of the fp16 inference (Ema) weights that sharded-Ema steps keep current only inside each rank's shard")
		.def_readwrite("dp_inference_stale", &Testbed::m_dp_inference_stale, "the inference (Ema) weights outside this rank's shard are old (see dp_sharded_ema); writable as a test hook")
		.def_readwrite("dp_state_stale", &Testbed::m_dp_state_stale, "sharded data-parallel steps ran since the last dp_gather_optimizer_state(): the fp32 master weights / Adam moments are current only inside this rank's shard.  optimizer steps outside a communicator and save_snapshot(include_optimizer_state=True) refuse such a state; dp_gather_optimizer_state() (with the communicator still live), reset_network() and load_snapshot() clear it.  Writable as a test hook (one-GPU tests cannot run a world of two)")
		.def_readwrite("render_sharded", &Testbed::m_render_sharded, "opt-in: under init_data_parallel render() / render_to_cpu() become COLLECTIVES (rows per rank, RCCL all-gather, every rank returns the whole frame) — set it on every rank and call render() on every rank with the same arguments.  Default False: render() is local and traces the whole frame, so one rank alone can render")
		.def_property_readonly("dp_comm_size", [](Testbed& t) { return t.m_dp_comm ? ngp_rccl_comm_size(t.m_dp_comm) : 0; }, "ranks of the RCCL communicator of init_data_parallel (ncclCommCount), 0 without one")
		.def_readonly("world_size", &Testbed::m_world_size)
		.def_readonly("rank", &Testbed::m_rank)
		.def_property("dp_march_behind_exchange", [](Testbed& t) { return t.m_dp_march_behind_exchange; }, &Testbed::set_dp_march_behind_exchange,
			"data-parallel step: hold the next step's march (second stream) until this step's gradients are final, so that it runs beside the RCCL exchange instead of beside the backward pass.  On by default when init_data_parallel is called with more than one rank; same samples either way")
		.def_property("strong_scaling", [](Testbed& t) { return t.m_dp_strong_scaling; }, &Testbed::set_dp_strong_scaling,
			"data-parallel batch split: True = train(B) back-propagates B / world per rank (the reference's convergence per step), False = B per rank (world x the global batch).  Set it on every rank between two steps; rays_per_batch re-adapts through the counter feedback within a few steps")
		.def("train_nerf_dp_begin", [](Testbed& t, uint32_t batch, bool get_loss, bool wait) { uint32_t c[2]; { py::gil_scoped_release rel; t.train_nerf_dp_begin(batch, c, get_loss, wait); } return py::make_tuple(c[0], c[1]); },
			py::arg("batch_size"), py::arg("get_loss_scalar") = false, py::arg("wait_for_counters") = true)
		.def("set_dp_counter_buffer", [](Testbed& t, uintptr_t p) { t.set_dp_counter_buffer((void*)p); })                 // 3 doubles on the device
		.def("stream_wait_counters", [](Testbed& t, uintptr_t stream) { t.stream_wait_counters((void*)stream); })         // hipStreamWaitEvent(stream, counters posted)
		.def("stream_wait_grid_gradients", [](Testbed& t, uintptr_t stream) { t.stream_wait_grid_gradients((void*)stream); })
		.def_property_readonly("n_mlp_params", [](Testbed& t) { return t.m_n_matrix_params; })   // the gradient vector is [MLP | grid]
		.def("train_nerf_dp_backward", &Testbed::train_nerf_dp_backward, py::call_guard<py::gil_scoped_release>(), py::arg("batch_size"), py::arg("measured_before_compaction"), py::arg("measured"),
			py::arg("get_loss_scalar") = false, py::arg("loss_sum") = 0.f)
		.def("train_nerf_dp_end", &Testbed::train_nerf_dp_end, py::call_guard<py::gil_scoped_release>())
		.def_property("render_camera_model", [](Testbed& t) { return (ECameraModel)t.m_render_camera_models.model; },          // python_api.cu:691-693
			[](Testbed& t, ECameraModel m) { t.m_render_camera_models.model = (int)m; })
		.def_property("camera_spherical_quadrilateral",
			[](Testbed& t) { SphericalQuadrilateral q; q.width = t.m_render_camera_models.sq_width; q.height = t.m_render_camera_models.sq_height; q.curvature = t.m_render_camera_models.sq_curvature; return q; },
			[](Testbed& t, const SphericalQuadrilateral& q) { t.m_render_camera_models.sq_width = q.width; t.m_render_camera_models.sq_height = q.height; t.m_render_camera_models.sq_curvature = q.curvature; })
		.def_property("camera_quadrilateral_hexahedron",
			[](Testbed& t) {
				QuadrilateralHexahedron h; Quadrilateral3D* q[2] = {&h.front, &h.back}; const float* src[2] = {t.m_render_camera_models.qh_front, t.m_render_camera_models.qh_back};
				for (int k = 0; k < 2; ++k) { Vec3* v[4] = {&q[k]->tl, &q[k]->tr, &q[k]->bl, &q[k]->br}; for (int i = 0; i < 4; ++i) { v[i]->x = src[k][3 * i]; v[i]->y = src[k][3 * i + 1]; v[i]->z = src[k][3 * i + 2]; } }
				return h;
			},
			[](Testbed& t, const QuadrilateralHexahedron& h) {
				const Quadrilateral3D* q[2] = {&h.front, &h.back}; float* dst[2] = {t.m_render_camera_models.qh_front, t.m_render_camera_models.qh_back};
				for (int k = 0; k < 2; ++k) { const Vec3 v[4] = {q[k]->tl, q[k]->tr, q[k]->bl, q[k]->br}; for (int i = 0; i < 4; ++i) { dst[k][3 * i] = v[i].x; dst[k][3 * i + 1] = v[i].y; dst[k][3 * i + 2] = v[i].z; } }
			})
		.def_readwrite("slice_plane_z", &Testbed::m_slice_plane_z)          // python_api.cu:662-666
		.def_readwrite("dof", &Testbed::m_aperture_size)
		.def_readwrite("aperture_size", &Testbed::m_aperture_size)
		.def_readwrite("autofocus", &Testbed::m_autofocus)
		.def_property("autofocus_target", [](Testbed& t) { py::array_t<float> a(3); for (int i = 0; i < 3; ++i) a.mutable_data()[i] = t.m_autofocus_target[i]; return a; },
			[](Testbed& t, const std::vector<float>& v) { if (v.size() != 3) throw std::runtime_error{"autofocus_target takes 3 floats"}; for (int i = 0; i < 3; ++i) t.m_autofocus_target[i] = v[i]; })
		.def_readwrite("prefetch_samples", &Testbed::m_enable_prefetch)
		.def_readwrite("x_row_index_mode", &Testbed::m_x_row_index_mode, "training step (base network family): True (default) = the loss kernel's compaction leaves, per kept sample, the index of its row in the uncompacted batch's encodings and the backward pass reads through it; False = it copies the 64-byte rows next to the compacted coordinates (rounds 1-5).  The same bits reach the backward pass")
		.def_readwrite("morton_grid_samples", &Testbed::m_morton_grid_samples, "occupancy-grid update: True (default) = the update's samples are written in Morton order of the cell each one's first try lands in (thread per cell, the index map inverted: csrc/density_grid.hip), so that the density pass and the splat touch shared lines from neighbouring lanes; False = the reference's order (sample i in slot i).  The same (position, index) pairs either way, hence the same grid bit for bit")
		.def_readwrite("separate_forward_pass", &Testbed::m_separate_forward)   // dev / test: also run the reference's second network pass (testbed_nerf.cu:3330)
		.def_readonly("prefetch_hits", &Testbed::m_prefetch_hits)
		.def_readonly("grid_prefetch_hits", &Testbed::m_grid_prefetch_hits)
		.def("training_prep_nerf", &Testbed::training_prep_nerf, py::call_guard<py::gil_scoped_release>(), py::arg("batch_size") = 0)
		.def("local_loss_sum", &Testbed::local_loss_sum)
		.def("gradients_ptr", [](Testbed& t) { return (uintptr_t)t.gradients(); })
		.def("params_ptr", [](Testbed& t) { return (uintptr_t)t.m_params.data(); })
		.def("sync", &Testbed::sync)
		.def("stream_ptr", [](Testbed& t) { return (uintptr_t)t.stream(); })  // hipStream_t of the training stream (torch.cuda.ExternalStream)
		.def("debug_pointers", [](Testbed& t) {  // dev tooling (tools/microbench.py): device addresses of the training inputs
				py::dict d;
				d["bitfield"] = (uintptr_t)t.m_nerf.density_grid_bitfield.data();
				d["metadata"] = (uintptr_t)t.m_nerf.training.dataset.metadata_gpu.data();
				d["xforms"] = (uintptr_t)t.m_nerf.training.transforms_gpu.data();
				d["n_images"] = t.m_nerf.training.n_images_for_training;
				d["rng_state"] = t.m_rng.state; d["rng_inc"] = t.m_rng.inc;
				d["rays_per_batch"] = t.m_nerf.training.counters_rgb.rays_per_batch;
				d["cone_angle_constant"] = t.m_nerf.cone_angle_constant;
				d["desc"] = (uintptr_t)t.m_desc_gpu.data();
				d["params"] = (uintptr_t)t.m_params.data();
				return d;
			})
		// test hook (tests/test_step_schedule_gpu.py): the sample positions / cell indices of the NEXT occupancy-grid update as stream B generated them ahead
		// (regenerate = false; empty when none is pending), or generated now in stream order from the same generator state (regenerate = true; state restored)
		.def("debug_grid_update_samples", [](Testbed& t, bool regenerate) {
				std::vector<float> pos; std::vector<uint32_t> idx; uint32_t step = 0;
				py::dict d;
				d["pending"] = t.debug_grid_update_samples(regenerate, pos, idx, step);
				d["step"] = step;
				py::array_t<float> p((py::ssize_t)pos.size()); py::array_t<uint32_t> i((py::ssize_t)idx.size());
				if (!pos.empty()) { memcpy(p.mutable_data(), pos.data(), pos.size() * 4); memcpy(i.mutable_data(), idx.data(), idx.size() * 4); }
				d["positions"] = p; d["indices"] = i;
				return d;
			}, py::arg("regenerate") = false)
		// test hooks (tests/test_baseline_configs_gpu.py): a stage-by-stage record of one product-path training step, and the scene as the kernels see it
		.def("debug_capture_next_step", &Testbed::debug_capture_next_step)
		.def("debug_captured", [](Testbed& t) {
				Testbed::StepCapture& c = t.m_capture;
				if (!c.valid) throw std::runtime_error{"debug_captured: no step was captured (call debug_capture_next_step() before frame())"};
				t.sync();
				auto bytes = [](const DeviceBuffer& b, size_t n) { py::array_t<uint8_t> a((py::ssize_t)n); if (n) b.copy_to_host(a.mutable_data(), n); return a; };
				auto u32 = [&](const DeviceBuffer& b, size_t n) { return bytes(b, n * 4).attr("view")("uint32"); };
				auto f32 = [&](const DeviceBuffer& b, size_t n) { return bytes(b, n * 4).attr("view")("float32"); };
				auto u16 = [&](const DeviceBuffer& b, size_t n) { return bytes(b, n * 2).attr("view")("uint16"); };
				py::dict d;
				d["step"] = c.step; d["R"] = c.R; d["max_inference"] = c.max_inference; d["n_rays_global"] = c.n_rays_global; d["ray_offset"] = c.ray_offset;
				d["target_batch_size"] = c.target_batch_size; d["rng_state"] = c.rng_state; d["rng_inc"] = c.rng_inc;
				d["params"] = u16(c.params, t.m_n_params);
				d["ray_indices"] = u32(c.ray_indices, c.R);
				d["rays"] = f32(c.rays, (size_t)c.R * 6);
				d["numsteps"] = u32(c.numsteps, (size_t)c.R * 2);
				d["coords"] = f32(c.coords, (size_t)c.max_inference * 7);
				d["gen_counters"] = u32(c.gen_counters, 2);                         // {rays kept, samples (may exceed max_inference)}
				d["density_grid_mean"] = f32(c.density_grid_mean, 1);
				d["bitfield"] = bytes(c.bitfield, c.bitfield.bytes());
				d["prefetch_hit"] = c.prefetch_hit;
				d["numsteps_compacted"] = u32(c.numsteps_compacted, (size_t)c.R * 2);
				d["coords_compacted"] = f32(c.coords_compacted, (size_t)c.target_batch_size * 7);   // before the roll-over
				d["dloss"] = u16(c.dloss, (size_t)c.target_batch_size * 4);                       // before the roll-over / rescale
				// buffers the step leaves behind (the prefetch of the next march is suppressed for a captured step)
				d["mlp_out"] = u16(t.debug_buffer("mlp_out"), (size_t)c.max_inference * 4);
				d["coords_compacted_rolled"] = f32(t.debug_buffer("coords_compacted"), (size_t)c.target_batch_size * 7);
				d["dloss_rolled"] = u16(t.debug_buffer("dloss"), (size_t)c.target_batch_size * 4);
				{   // the compacted batch's encoding rows (gathered through the row index when the step kept the rows in the uncompacted batch)
					const std::vector<uint16_t> rows = t.debug_x_saved((size_t)c.target_batch_size);
					py::array_t<uint16_t> xa((py::ssize_t)rows.size());
					if (!rows.empty()) memcpy(xa.mutable_data(), rows.data(), rows.size() * 2);
					d["x_saved"] = xa;
				}
				d["grads"] = u16(t.debug_buffer("grads"), t.m_n_params);
				if (t.m_nerf.training.optimize_extrinsics) d["coords_gradient"] = f32(t.debug_buffer("coords_gradient"), (size_t)c.target_batch_size * 6);   // dL/d(pos, dir) of the rolled-over batch
				d["loss"] = f32(t.m_nerf.training.counters_rgb.loss, c.n_rays_global);
				d["measured_batch_size"] = t.m_nerf.training.counters_rgb.measured_batch_size;
				d["measured_batch_size_before_compaction"] = t.m_nerf.training.counters_rgb.measured_batch_size_before_compaction;
				return d;
			})
		.def("debug_params", [](Testbed& t, const std::string& which) {   // fp16 bits of the "training" or "inference" (EMA) weights
				if (which == "inference") t.require_inference_params("debug_params('inference')", false);
				const DeviceBuffer& b = which == "inference" ? t.m_inference_params : t.m_params;
				t.sync();
				py::array_t<uint16_t> a((py::ssize_t)t.m_n_params);
				if (t.m_n_params) b.copy_to_host(a.mutable_data(), t.m_n_params * 2);
				return a;
			}, py::arg("which") = "training")
		.def("debug_scene", [](Testbed& t) {
				py::dict d;
				const NerfTraining& tr = t.m_nerf.training;
				const size_t n = (size_t)tr.n_images_for_training;
				py::array_t<uint8_t> md((py::ssize_t)(n * sizeof(NgpImageMeta))), xf((py::ssize_t)(n * sizeof(NgpXForm)));
				if (n) { memcpy(md.mutable_data(), tr.dataset.metadata.data(), n * sizeof(NgpImageMeta)); memcpy(xf.mutable_data(), tr.transforms.data(), n * sizeof(NgpXForm)); }
				d["metadata"] = md; d["xforms"] = xf;                                // NgpImageMeta / NgpXForm records (pixel pointers are DEVICE pointers)
				py::array_t<uint8_t> bf((py::ssize_t)t.m_nerf.density_grid_bitfield.bytes());
				if (bf.size()) t.m_nerf.density_grid_bitfield.copy_to_host(bf.mutable_data(), (size_t)bf.size());
				d["bitfield"] = bf;
				py::array_t<float> grid((py::ssize_t)(t.m_nerf.density_grid.bytes() / 4));
				if (grid.size()) t.m_nerf.density_grid.copy_to_host(grid.mutable_data(), (size_t)grid.size() * 4);
				d["density_grid"] = grid;
				d["aabb"] = py::make_tuple(std::vector<float>(t.m_aabb.min, t.m_aabb.min + 3), std::vector<float>(t.m_aabb.max, t.m_aabb.max + 3));
				d["n_cascades"] = t.m_nerf.max_cascade + 1;
				d["cone_angle_constant"] = t.m_nerf.cone_angle_constant;
				d["near_distance"] = tr.near_distance;
				d["error_map_res"] = std::vector<int>{tr.error_map_res[0], tr.error_map_res[1]};
				d["rgb_activation"] = (int)t.m_nerf.rgb_activation; d["density_activation"] = (int)t.m_nerf.density_activation;
				d["loss_type"] = (int)tr.loss_type;
				py::array_t<uint8_t> desc((py::ssize_t)sizeof(NgpNetDesc)); memcpy(desc.mutable_data(), &t.m_desc, sizeof(NgpNetDesc));
				d["desc"] = desc;
				return d;
			})
		// live per-kernel timing with HIP events on the launch stream (bench.py roofline numbers)
		.def("set_profiling", [](Testbed& t, bool on, const std::vector<std::string>& only, uint32_t every) {
				// only: names as in profile(); empty = all kinds.  Every bracketed launch group costs two event records on the stream.
				static const char* names[Testbed::PK_COUNT] = {"generate_training_samples", "nerf_inference", "compute_loss", "nerf_forward", "nerf_backward", "optimizer_step", "density_grid_prep", "grad_exchange", "param_gather"};
				uint32_t mask = only.empty() ? ~0u : 0u;
				for (const auto& n : only) {
					bool found = false;
					for (int k = 0; k < Testbed::PK_COUNT; ++k) if (n == names[k]) { mask |= 1u << k; found = true; }
					if (!found) throw std::invalid_argument{"set_profiling: unknown kernel group '" + n + "'"};
				}
				t.m_profile_enabled = on; t.m_profile_mask = mask; t.m_profile_every = every ? every : 1u;   // every: bracket the launches of every n-th training step only
			}, py::arg("on"), py::arg("only") = std::vector<std::string>{}, py::arg("every") = 1u)
		.def("reset_profile", &Testbed::reset_profile)
		.def_property("network_pass", [](Testbed& t) { return std::string(t.m_network_pass == Testbed::ENetworkPass::Auto ? "auto" : t.m_network_pass == Testbed::ENetworkPass::Fused ? "fused" : "two_kernel"); },
			[](Testbed& t, const std::string& v) {
				if (v == "auto") t.m_network_pass = Testbed::ENetworkPass::Auto; else if (v == "fused") t.m_network_pass = Testbed::ENetworkPass::Fused; else if (v == "two_kernel") t.m_network_pass = Testbed::ENetworkPass::TwoKernel;
				else throw std::runtime_error{"network_pass: 'auto', 'fused' or 'two_kernel'"};
			}, "organisation of the network pass over a training batch: 'fused' (hash gathers inside the MLP kernel), 'two_kernel' (XCD-affine encode into level planes + MLP kernel), or 'auto' (default): both are timed on "
			   "this workload for a few steps (HIP events) and the faster one runs; same bits either way.  The reference has one path (src/testbed_nerf.cu:3256, nerf_network.h:103-137)")
		.def_property_readonly("network_pass_report", [](Testbed& t) {
				const Testbed::NetworkPassTuner& u = t.m_pass_tuner;
				py::dict d;
				const bool forced = t.m_network_pass != Testbed::ENetworkPass::Auto;
				const Testbed::ENetworkPass running = forced ? t.m_network_pass : u.chosen;
				d["policy"] = t.m_network_pass == Testbed::ENetworkPass::Auto ? "auto" : "forced";
				d["running"] = running == Testbed::ENetworkPass::TwoKernel ? "two_kernel" : "fused";
				d["calibrations"] = u.n_calibrations; d["last_calibration_step"] = u.last_calibration_step;
				d["fused_us"] = u.last_us[0]; d["two_kernel_us"] = u.last_us[1];
				return d;
			}, "what network_pass = 'auto' measured last (medians of the bracketed launches, microseconds) and what is running")
		.def_readwrite("trace_sync", &Testbed::m_trace_sync, "debugging aid: drain both streams behind every launch group and name it on stderr")
		.def_readwrite("render_trace", &Testbed::m_render_trace, "debugging aid: the tracers' pass structure (alive rays, steps per pass) on stderr")
		.def_readwrite("async_training_steps", &Testbed::m_async_training_steps, "frame() does not drain the stream after the training step (the reference does, testbed.cu:2570): the next step's launches queue behind this one's optimizer instead of after an idle gap.  Everything the API reads afterwards is ordered by the same stream; call sync() before touching device buffers from another stream.")
		.def("profile", [](Testbed& t) {
				t.sync();
				t.profile_collect();
				static const char* names[Testbed::PK_COUNT] = {"generate_training_samples", "nerf_inference", "compute_loss", "nerf_forward", "nerf_backward", "optimizer_step", "density_grid_prep", "grad_exchange", "param_gather"};
				py::dict d;
				for (int k = 0; k < Testbed::PK_COUNT; ++k) {
					py::dict e;
					e["ms"] = t.m_prof[k].ms; e["launches"] = t.m_prof[k].launches; e["units"] = t.m_prof[k].units;
					d[names[k]] = e;
				}
				return d;
			})
		.def_readwrite("shall_train", &Testbed::m_train)
		.def_readwrite("shall_train_encoding", &Testbed::m_train_encoding)      // python_api.cu:656-657
		.def_readwrite("shall_train_network", &Testbed::m_train_network)
		.def_readwrite("render_groundtruth", &Testbed::m_render_ground_truth)   // GUI overlays and window state below: stored, read back, consumed by nothing headless
		.def_property("groundtruth_render_mode", [](Testbed& t) { return (EGroundTruthRenderMode)t.m_ground_truth_render_mode; }, [](Testbed& t, EGroundTruthRenderMode v) { t.m_ground_truth_render_mode = (int)v; })
		.def_readwrite("dynamic_res_target_fps", &Testbed::m_dynamic_res_target_fps)
		.def_readwrite("fixed_res_factor", &Testbed::m_fixed_res_factor)
		.def_readwrite("floor_enable", &Testbed::m_floor_enable)
		.def_readwrite("display_gui", &Testbed::m_imgui_enabled)
		.def_readwrite("visualize_unit_cube", &Testbed::m_visualize_unit_cube)
		.def_readwrite("visualized_dimension", &Testbed::m_visualized_dimension)
		.def_readwrite("visualized_layer", &Testbed::m_visualized_layer)
		.def_readwrite("dlss_sharpening", &Testbed::m_dlss_sharpening)
		.def_property("dlss", [](Testbed& t) { return t.m_dlss; }, [](Testbed& t, bool v) { if (v) throw std::runtime_error{"DLSS requires a Window to be initialized via `init_window`."}; t.m_dlss = false; })   // python_api.cu:713-727
		.def_readonly("bounding_radius", &Testbed::m_bounding_radius)
		.def_property("render_aabb_to_local", [](Testbed& t) { py::array_t<float> a({3, 3}); for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) a.mutable_at(r, c) = t.m_render_aabb_to_local[3 * c + r]; return a; },
			[](Testbed& t, const py::array_t<float, py::array::c_style | py::array::forcecast>& a) { if (a.ndim() != 2 || a.shape(0) != 3 || a.shape(1) != 3) throw std::runtime_error{"expected a 3x3 matrix"}; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) t.m_render_aabb_to_local[3 * c + r] = a.at(r, c); })
		.def_property("fov_xy", [](Testbed& t) { float f[2]; t.fov_xy(f); py::array_t<float> a(2); a.mutable_data()[0] = f[0]; a.mutable_data()[1] = f[1]; return a; },
			[](Testbed& t, const std::vector<float>& v) { if (v.size() != 2) throw std::runtime_error{"fov_xy takes 2 floats"}; t.set_fov_xy(v.data()); })
		.def_property("sun_dir", [](Testbed& t) { return vec3_to_py(t.m_sun_dir); }, [](Testbed& t, const py::object& v) { t.m_sun_dir = vec3_from_py(v); })
		.def_property("look_at", [](Testbed& t) { return vec3_to_py(t.look_at()); }, [](Testbed& t, const py::object& v) { t.set_look_at(vec3_from_py(v)); })
		.def_property("view_dir", [](Testbed& t) { return vec3_to_py(t.view_dir()); }, [](Testbed& t, const py::object& v) { t.set_view_dir(vec3_from_py(v)); })
		.def_property("parallax_shift", [](Testbed& t) { return std::vector<float>(t.m_parallax_shift, t.m_parallax_shift + 3); }, [](Testbed& t, const std::vector<float>& v) { for (int k = 0; k < 3; ++k) t.m_parallax_shift[k] = v.at(k); })
		.def_readwrite("exposure", &Testbed::m_exposure)
		.def_readwrite("snap_to_pixel_centers", &Testbed::m_snap_to_pixel_centers)
		.def_readwrite("fov_axis", &Testbed::m_fov_axis)
		.def_property("fov", &Testbed::fov, &Testbed::set_fov)
		.def_readwrite("color_space", &Testbed::m_color_space)
		.def_readwrite("zoom", &Testbed::m_zoom)
		.def_property("scale", &Testbed::scale, &Testbed::set_scale)            // python_api.cu:669 (moves the camera along its view ray, testbed.cu:231-235)
		.def_readwrite("seed", &Testbed::m_seed)
		.def_readwrite("training_batch_size", &Testbed::m_training_batch_size)
		.def_readwrite("max_level_rand_training", &Testbed::m_max_level_rand_training)
		.def_readwrite("render_near_distance", &Testbed::m_render_near_distance)
		.def_property("tonemap_curve", [](Testbed& t) { return t.m_windowless_render_surface.tonemap_curve; }, [](Testbed& t, ETonemapCurve c) { t.m_windowless_render_surface.tonemap_curve = c; })
		.def_property("background_color", [](Testbed& t) { return std::vector<float>(t.m_background_color, t.m_background_color + 4); },
			[](Testbed& t, const std::vector<float>& c) { if (c.size() != 4) throw std::runtime_error{"background_color needs 4 components"}; for (int i = 0; i < 4; ++i) t.m_background_color[i] = c[i]; })
		.def_property("screen_center", [](Testbed& t) { return std::vector<float>(t.m_screen_center, t.m_screen_center + 2); },
			[](Testbed& t, const std::vector<float>& c) { t.m_screen_center[0] = c.at(0); t.m_screen_center[1] = c.at(1); })
		.def_property("camera_matrix", [](Testbed& t) { return mat34_to_py(t.m_camera); },
			[](Testbed& t, const py::array_t<float, py::array::c_style | py::array::forcecast>& cam) { t.m_camera = mat34_from_py(cam); })
		.def_property_readonly("training_step", &Testbed::training_step)
		.def_property_readonly("loss", &Testbed::loss)
		.def_property_readonly("learning_rate", [](Testbed& t) { return t.m_learning_rate; })
		.def_property_readonly("render_ms", [](Testbed& t) { return t.m_stats.render_ms; })
		.def_property_readonly("training_ms", [](Testbed& t) { return t.m_stats.training_ms; })
		.def_property_readonly("training_prep_ms", [](Testbed& t) { return t.m_stats.training_prep_ms; })
		.def_property_readonly("render_samples_evaluated", [](Testbed& t) { return t.m_render_samples_evaluated; })
		.def_property_readonly("aabb", [](Testbed& t) { return py::make_tuple(std::vector<float>(t.m_aabb.min, t.m_aabb.min + 3), std::vector<float>(t.m_aabb.max, t.m_aabb.max + 3)); })
		// names scripts/run.py and the add-on touch on the NeRF path (python_api.cu:650-732); state that only the GUI consumes is stored as is
		.def_readwrite("camera_smoothing", &Testbed::m_camera_smoothing)
		.def_readwrite("loop_animation", &Testbed::m_loop_animation)
		.def_readwrite("dynamic_res", &Testbed::m_dynamic_res)
		.def_property("render_aabb", [](Testbed& t) { return BoundingBox(Vec3{t.m_render_aabb.min[0], t.m_render_aabb.min[1], t.m_render_aabb.min[2]}, Vec3{t.m_render_aabb.max[0], t.m_render_aabb.max[1], t.m_render_aabb.max[2]}); },
			[](Testbed& t, const BoundingBox& b) { t.m_render_aabb = b.pod(); })
		.def_property_readonly("raw_aabb", [](Testbed& t) { return BoundingBox(Vec3{t.m_raw_aabb.min[0], t.m_raw_aabb.min[1], t.m_raw_aabb.min[2]}, Vec3{t.m_raw_aabb.max[0], t.m_raw_aabb.max[1], t.m_raw_aabb.max[2]}); })
		.def_property("up_dir", [](Testbed& t) { return vec3_to_py(t.m_up_dir); }, [](Testbed& t, const py::object& v) { t.m_up_dir = vec3_from_py(v); })
		.def_readwrite("render_mode", &Testbed::m_render_mode)   // python_api.cu:660: every ERenderMode of the stock tracer is built
		.def_property("render_masks", [](Testbed& t) { std::vector<Mask3D> v; for (const NgpMask3D& p : t.m_render_masks) { Mask3D m; m.pod = p; v.push_back(m); } return v; },   // python_api.cu:694
			[](Testbed& t, const std::vector<Mask3D>& v) { t.m_render_masks.clear(); for (const Mask3D& m : v) t.m_render_masks.push_back(m.pod); })
		// extensions (no counterpart in the reference's module): the two trainable 2-D buffers as arrays, so that scripts and tests can inspect / seed them
		.def("get_envmap", [](Testbed& t) { t.sync(); py::array_t<float> a({(size_t)t.m_envmap.resolution[1], (size_t)t.m_envmap.resolution[0], (size_t)4});
				if (t.m_envmap.n_params()) t.m_envmap.params.copy_to_host(a.mutable_data(), t.m_envmap.n_params() * 4); return a; })
		.def("get_distortion_map", [](Testbed& t) { t.sync(); py::array_t<float> a({(size_t)t.m_distortion.resolution[1], (size_t)t.m_distortion.resolution[0], (size_t)2});
				if (t.m_distortion.n_params()) t.m_distortion.params.copy_to_host(a.mutable_data(), t.m_distortion.n_params() * 4); return a; })
		.def("set_distortion_map", [](Testbed& t, py::array_t<float, py::array::c_style | py::array::forcecast> a) {
				if ((size_t)a.size() != t.m_distortion.n_params()) throw std::runtime_error{"set_distortion_map: expected resolution[1] x resolution[0] x 2 floats"};
				t.sync(); t.m_distortion.set_params(a.data(), (size_t)a.size()); t.invalidate_training_inputs(); }, py::arg("map"))
		.def_property("quilting_dims", [](Testbed& t) { return std::vector<int>{t.m_quilting_dims[0], t.m_quilting_dims[1]}; },   // testbed.h:549 (GUI-only in the reference)
			[](Testbed& t, const std::vector<int>& v) { if (v.size() != 2 || v[0] < 1 || v[1] < 1) throw std::runtime_error{"quilting_dims: two positive integers"}; t.m_quilting_dims[0] = v[0]; t.m_quilting_dims[1] = v[1]; })
		.def("set_camera_to_training_view", &Testbed::set_camera_to_training_view, py::arg("trainview"))
		.def("clear_training_data", [](Testbed& t) { t.m_training_data_available = false; t.m_nerf.training.n_images_for_training = 0; })
		.def_readonly("nerf", &Testbed::m_nerf)
		.def_readonly("image", &Testbed::m_image)                                // python_api.cu:707, 875-886
		.def_readonly("sdf", &Testbed::m_sdf);

	py::class_<Testbed::ImageState> image(testbed, "Image");
	image
		.def_property_readonly("training", [](Testbed::ImageState& i) -> Testbed::ImageState& { return i; }, py::return_value_policy::reference_internal)   // .training.snap_to_pixel_centers / .linear_colors live on the same record
		.def_readwrite("snap_to_pixel_centers", &Testbed::ImageState::snap_to_pixel_centers)
		.def_readwrite("linear_colors", &Testbed::ImageState::linear_colors)
		.def_property("random_mode", [](Testbed::ImageState& i) { return i.stratified ? ERandomMode::Stratified : ERandomMode::Random; },
			[](Testbed::ImageState& i, ERandomMode m) { if (m != ERandomMode::Stratified && m != ERandomMode::Random) throw std::runtime_error{"Image.random_mode: Random and Stratified are built (Halton / Sobol are not)"}; i.stratified = m == ERandomMode::Stratified; })
		.def_property("pos", [](Testbed::ImageState& i) { return std::vector<float>{i.pos[0], i.pos[1]}; }, [](Testbed::ImageState& i, const std::vector<float>& v) { i.pos[0] = v.at(0); i.pos[1] = v.at(1); });
	py::class_<Testbed::SdfState>(testbed, "Sdf")
		.def_readwrite("mesh_scale", &Testbed::SdfState::mesh_scale);

	py::class_<Nerf> nerf(testbed, "Nerf");
	nerf
		.def_readonly("training", &Nerf::training)
		.def_readwrite("rgb_activation", &Nerf::rgb_activation)
		.def_readwrite("density_activation", &Nerf::density_activation)
		.def_readwrite("sharpen", &Nerf::sharpen)
		.def_readwrite("render_with_lens_distortion", &Nerf::render_with_lens_distortion)
		.def_readwrite("render_min_transmittance", &Nerf::render_min_transmittance)
		.def_readwrite("rendering_min_transmittance", &Nerf::render_min_transmittance)      // the name scripts/run.py uses (python_api.cu:751)
		.def_readwrite("render_with_camera_distortion", &Nerf::render_with_lens_distortion)  // legacy name (python_api.cu:747)
		.def_readwrite("render_max_steps_per_pass", &Nerf::render_max_steps_per_pass)
		.def_readwrite("render_n_streams", &Nerf::render_n_streams)
		.def_readwrite("cone_angle_constant", &Nerf::cone_angle_constant)
		.def_readwrite("show_accel", &Nerf::show_accel)
		.def_readwrite("render_tile_order", &Nerf::render_tile_order)
		.def_readwrite("render_max_skips_per_pass", &Nerf::render_max_skips_per_pass)
		.def_readwrite("render_pass_samples_factor", &Nerf::render_pass_samples_factor)
		.def_readwrite("render_fused_compaction", &Nerf::render_fused_compaction)
		.def_readwrite("render_fused_network", &Nerf::render_fused_network)
		.def_property("light_dir", [](Nerf& n) { return vec3_to_py(n.light_dir); }, [](Nerf& n, const py::object& v) { n.light_dir = vec3_from_py(v); })   // testbed.h:712 (GUI slider in the reference)
		.def_readwrite("extra_dim_idx_for_inference", &Nerf::extra_dim_idx_for_inference)                                                                     // testbed.h:713
		.def_readwrite("visualize_cameras", &Nerf::visualize_cameras)
		.def_readwrite("glow_y_cutoff", &Nerf::glow_y_cutoff)
		.def_readwrite("glow_mode", &Nerf::glow_mode)
		.def_property("render_lens", [](Nerf& n) { return PyLens::from(n.render_lens_proxy); }, [](Nerf& n, const PyLens& l) { l.to(n.render_lens_proxy); })          // python_api.cu:750-751
		.def_property("render_distortion", [](Nerf& n) { return PyLens::from(n.render_lens_proxy); }, [](Nerf& n, const PyLens& l) { l.to(n.render_lens_proxy); })   // legacy name
		.def_readonly("max_cascade", &Nerf::max_cascade)
		.def("density_grid_bitfield", [](Nerf& n) {
				py::array_t<uint8_t> a((py::ssize_t)n.density_grid_bitfield.bytes());
				n.density_grid_bitfield.copy_to_host(a.mutable_data(), n.density_grid_bitfield.bytes());
				return a;
			});

	py::class_<NerfTraining> training(nerf, "Training");
	training
		.def_readwrite("random_bg_color", &NerfTraining::random_bg_color)
		.def_readwrite("linear_colors", &NerfTraining::linear_colors)
		.def_readwrite("loss", &NerfTraining::loss_type)
		.def_readwrite("loss_type", &NerfTraining::loss_type)
		.def_property_readonly("n_images", [](NerfTraining& t) { return t.dataset.n_images; })
		.def_property_readonly("aabb_scale", [](NerfTraining& t) { return t.dataset.aabb_scale; })
		.def_property_readonly("is_hdr", [](NerfTraining& t) { return t.dataset.is_hdr; })
		.def_property_readonly("from_mitsuba", [](NerfTraining& t) { return t.dataset.from_mitsuba; })
		.def_property_readonly("paths", [](NerfTraining& t) { return t.dataset.paths; })
		.def_property_readonly("scale", [](NerfTraining& t) { return t.dataset.scale; })
		.def_property_readonly("offset", [](NerfTraining& t) { return vec3_to_py(t.dataset.offset); })
		.def_readwrite("snap_to_pixel_centers", &NerfTraining::snap_to_pixel_centers)
		.def_readwrite("near_distance", &NerfTraining::near_distance)
		.def_readwrite("optimize_exposure", &NerfTraining::optimize_exposure)                   // python_api.cu:813
		.def_readwrite("optimize_extrinsics", &NerfTraining::optimize_extrinsics)               // python_api.cu:811
		.def_readwrite("optimize_extra_dims", &NerfTraining::optimize_extra_dims)               // python_api.cu:812: per-image latent codes (n_extra_learnable_dims of the dataset)
		.def("get_extra_dims", [](NerfTraining& t) {   // the latent codes / light directions as the network sees them: (n_images, n_extra_dims) fp32 (extension: tests, scripts)
				const uint32_t ne = t.dataset.n_extra_dims();
				py::array_t<float> a({(py::ssize_t)t.dataset.n_images, (py::ssize_t)ne});
				if (ne && t.extra_dims_gpu.bytes() >= t.dataset.n_images * ne * 4) t.extra_dims_gpu.copy_to_host(a.mutable_data(), t.dataset.n_images * ne * 4);
				return a;
			})
		.def_readwrite("optimize_distortion", &NerfTraining::optimize_distortion)
		.def_readwrite("train_envmap", &NerfTraining::train_envmap)   // testbed.h:656 (the reference flips it from its GUI)
		.def_readwrite("optimize_focal_length", &NerfTraining::optimize_focal_length)           // python_api.cu:815: accepted; trains nothing, like the reference (testbed.h note)
		.def("reset_camera_extrinsics", [](NerfTraining& t) { t.reset_camera_extrinsics(); })   // testbed_nerf.cu:2543-2555 (not bound by the reference's pyngp; GUI button there)
		.def("_cam_offsets", [](NerfTraining& t) {   // test hook: (positions [n][3], angle-axis rotations [n][3], optimizer steps [n]) of optimize_extrinsics
				const size_t n = t.cam_pos_offset.size();
				py::array_t<float> pos({(py::ssize_t)n, (py::ssize_t)3}), rot({(py::ssize_t)n, (py::ssize_t)3});
				py::array_t<uint32_t> it((py::ssize_t)n);
				for (size_t i = 0; i < n; ++i) {
					for (int c = 0; c < 3; ++c) { pos.mutable_at(i, c) = t.cam_pos_offset[i].variable[c]; rot.mutable_at(i, c) = i < t.cam_rot_offset.size() ? t.cam_rot_offset[i].variable[c] : 0.f; }
					it.mutable_at(i) = t.cam_pos_offset[i].iter;
				}
				return py::make_tuple(pos, rot, it);
			})
		.def("_cam_gradients", [](NerfTraining& t) {   // test hook: the per-image gradients the last camera update consumed (summed over its window and over the ranks)
				const size_t n = t.cam_pos_gradient.size() / 3;
				py::array_t<float> pos({(py::ssize_t)n, (py::ssize_t)3}), rot({(py::ssize_t)n, (py::ssize_t)3});
				if (n) { memcpy(pos.mutable_data(), t.cam_pos_gradient.data(), n * 12); memcpy(rot.mutable_data(), t.cam_rot_gradient.data(), n * 12); }
				return py::make_tuple(pos, rot);
			})
		.def_readwrite("n_steps_between_cam_updates", &NerfTraining::n_steps_between_cam_updates)
		.def_readwrite("include_sharpness_in_error", &NerfTraining::include_sharpness_in_error)
		.def("get_sharpness_data", [](NerfTraining& t) {   // (n_images, 72, 128) tile sharpness as the loss kernel reads it (extension: tests)
				t.dataset.update_sharpness();
				py::array_t<float> a({(py::ssize_t)t.dataset.n_images, (py::ssize_t)t.dataset.sharpness_resolution[1], (py::ssize_t)t.dataset.sharpness_resolution[0]});
				if (t.dataset.n_images) t.dataset.sharpness_data.copy_to_host(a.mutable_data(), (size_t)a.size() * 4);
				return a;
			})
		.def_readwrite("extrinsic_l2_reg", &NerfTraining::extrinsic_l2_reg)
		.def_readwrite("extrinsic_learning_rate", &NerfTraining::extrinsic_learning_rate)
		.def_readwrite("intrinsic_l2_reg", &NerfTraining::intrinsic_l2_reg)
		.def_property_readonly("dataset", py::cpp_function([](NerfTraining& t) { return PyNerfDataset{&t.dataset}; }, py::keep_alive<0, 1>()))   // python_api.cu:830
		.def_property_readonly("transforms", [](NerfTraining& t) { py::list l; for (const NgpXForm& x : t.transforms) { Mat34 a, b; memcpy(a.m, x.start, 48); memcpy(b.m, x.end, 48); l.append(py::make_tuple(mat34_to_py(a), mat34_to_py(b))); } return l; })
		.def_readwrite("exposure_l2_reg", &NerfTraining::exposure_l2_reg)                       // python_api.cu:827
		.def("get_camera_exposures", [](NerfTraining& t) {                                      // [n_images][3] log2 exposures as the loss kernel sees them
				py::array_t<float> a({(py::ssize_t)t.dataset.n_images, (py::ssize_t)3});
				if (a.size()) t.cam_exposure_gpu.copy_to_host(a.mutable_data(), (size_t)a.size() * 4);
				return a;
			})
		.def_readwrite("depth_loss_type", &NerfTraining::depth_loss_type)                       // python_api.cu:809
		.def_readwrite("depth_supervision_lambda", &NerfTraining::depth_supervision_lambda)     // python_api.cu:828
		.def_readwrite("sample_focal_plane_proportional_to_error", &NerfTraining::sample_focal_plane_proportional_to_error)   // python_api.cu:817
		.def_readwrite("sample_image_proportional_to_error", &NerfTraining::sample_image_proportional_to_error)               // python_api.cu:818
		.def_readwrite("n_steps_between_error_map_updates", &NerfTraining::n_steps_between_error_map_updates)
		.def_readonly("n_steps_since_error_map_update", &NerfTraining::n_steps_since_error_map_update)
		.def_property_readonly("is_cdf_valid", [](NerfTraining& t) { return t.is_cdf_valid; })
		.def("get_error_map", [](NerfTraining& t) {               // [n_images][res_y][res_x] fp32, as accumulated since the last CDF update
				py::array_t<float> a({(py::ssize_t)t.dataset.n_images, (py::ssize_t)t.error_map_res[1], (py::ssize_t)t.error_map_res[0]});
				if (a.size()) t.error_map_data.copy_to_host(a.mutable_data(), (size_t)a.size() * 4);
				return a;
			})
		.def("get_error_map_cdfs", [](NerfTraining& t) {          // (cdf_x_cond_y [n][h][w], cdf_y [n][h], cdf_img [n], pmf_img [n])
				if (!t.is_cdf_valid) throw std::runtime_error{"the error-map CDFs have not been built yet"};
				const py::ssize_t n = (py::ssize_t)t.dataset.n_images, h = t.cdf_res[1], w = t.cdf_res[0];
				py::array_t<float> x({n, h, w}), y({n, h}), im({n}), pm({n});
				t.cdf_x_cond_y.copy_to_host(x.mutable_data(), (size_t)x.size() * 4);
				t.cdf_y.copy_to_host(y.mutable_data(), (size_t)y.size() * 4);
				t.cdf_img.copy_to_host(im.mutable_data(), (size_t)im.size() * 4);
				for (py::ssize_t i = 0; i < n; ++i) pm.mutable_data()[i] = t.pmf_img_cpu[(size_t)i];
				return py::make_tuple(x, y, im, pm);
			})
		.def_readwrite("density_grid_decay", &NerfTraining::density_grid_decay)
		.def_readwrite("n_images_for_training", &NerfTraining::n_images_for_training)
		.def_property_readonly("rays_per_batch", [](NerfTraining& t) { return t.counters_rgb.rays_per_batch; })
		.def_property_readonly("measured_batch_size", [](NerfTraining& t) { return t.counters_rgb.measured_batch_size; })
		.def_property_readonly("measured_batch_size_before_compaction", [](NerfTraining& t) { return t.counters_rgb.measured_batch_size_before_compaction; })
		.def_property_readonly("dataset_scale", [](NerfTraining& t) { return t.dataset.scale; })
		.def("set_dataset_transform", [](NerfTraining& t, float scale, const std::vector<float>& offset) {
				t.dataset.scale = scale; t.dataset.offset.x = offset.at(0); t.dataset.offset.y = offset.at(1); t.dataset.offset.z = offset.at(2);
			}, py::arg("scale"), py::arg("offset"), "the `scale` / `offset` keys of transforms.json (nerf_loader.cu:472-474, 499-504)")
		.def("set_image", [](NerfTraining& t, int frame_idx, const py::array_t<float, py::array::c_style | py::array::forcecast>& img,
		                     const py::array_t<float, py::array::c_style | py::array::forcecast>& depth_img, float depth_scale) {
				auto b = img.request();
				if (b.ndim != 3 || b.shape[2] != 4) throw std::runtime_error{"image should be (H,W,C) where C=4"};
				auto d = depth_img.request();
				const float* depth = nullptr;
				if (d.size > 0) {
					if (d.size != b.shape[0] * b.shape[1]) throw std::runtime_error{"depth image should be (H,W)"};
					depth = (const float*)d.ptr;
				}
				t.set_image(frame_idx, (int)b.shape[1], (int)b.shape[0], (const float*)b.ptr, depth, depth_scale);
			}, py::arg("frame_idx"), py::arg("img"), py::arg("depth_img") = py::array_t<float>(), py::arg("depth_scale") = -1.f)   // python_api.cu:53-72, 786-791
		.def("set_image_rgba8", [](NerfTraining& t, int frame_idx, const py::array_t<uint8_t, py::array::c_style | py::array::forcecast>& img) {
				auto b = img.request();
				if (b.ndim != 3 || b.shape[2] != 4) throw std::runtime_error{"image should be (H,W,C) where C=4"};
				t.set_image_rgba8(frame_idx, (int)b.shape[1], (int)b.shape[0], (const uint8_t*)b.ptr);
			}, py::arg("frame_idx"), py::arg("img"))
		.def("set_camera_extrinsics", [](NerfTraining& t, int frame_idx, const py::array_t<float, py::array::c_style | py::array::forcecast>& m, bool convert_to_ngp) {
				t.set_camera_extrinsics(frame_idx, mat34_from_py(m), convert_to_ngp);
			}, py::arg("frame_idx"), py::arg("camera_to_world"), py::arg("convert_to_ngp") = true)
		.def("get_camera_extrinsics", [](NerfTraining& t, int frame_idx) { return mat34_to_py(t.get_camera_extrinsics(frame_idx)); }, py::arg("frame_idx"))
		.def("get_image_metadata", [](NerfTraining& t, int i) {   // what the kernels see for training image i (TrainingImageMetadata, nerf_loader.h:30-45)
				if (i < 0 || (size_t)i >= t.dataset.n_images) throw std::runtime_error{"Invalid frame index"};
				const NgpImageMeta& m = t.dataset.metadata[i];
				py::dict d;
				d["resolution"] = std::vector<int>{m.res[0], m.res[1]};
				d["focal_length"] = std::vector<float>{m.focal_length[0], m.focal_length[1]};
				d["principal_point"] = std::vector<float>{m.principal_point[0], m.principal_point[1]};
				d["rolling_shutter"] = std::vector<float>(m.rolling_shutter, m.rolling_shutter + 4);
				d["lens_mode"] = m.lens_mode; d["lens_params"] = std::vector<float>(m.lens_params, m.lens_params + 7);
				d["image_data_type"] = m.image_data_type;
				if (m.depth) d["has_depth"] = true;    // (only present when set, so that records of the file route and the in-memory route compare equal)
				if (m.rays) d["has_rays"] = true;
				return d;
			}, py::arg("frame_idx"))
		.def("get_image_pixels", [](NerfTraining& t, int i) -> py::object {   // device copy of a training image as stored: uint8 / float16 / float32 (H, W, 4)
				if (i < 0 || (size_t)i >= t.dataset.n_images) throw std::runtime_error{"Invalid frame index"};
				const NgpImageMeta& m = t.dataset.metadata[i];
				const size_t px = (size_t)m.res[0] * m.res[1];
				if (m.image_data_type == 1) { py::array_t<uint8_t> a({m.res[1], m.res[0], 4}); t.dataset.pixelmemory[i].copy_to_host(a.mutable_data(), px * 4); return a; }
				if (m.image_data_type == 2) { py::array_t<uint16_t> a({m.res[1], m.res[0], 4}); t.dataset.pixelmemory[i].copy_to_host(a.mutable_data(), px * 8); return a.attr("view")("float16"); }
				py::array_t<float> a({m.res[1], m.res[0], 4}); t.dataset.pixelmemory[i].copy_to_host(a.mutable_data(), px * 16); return a;
			}, py::arg("frame_idx"))
		.def("get_image_rgba8", [](NerfTraining& t, int i) {      // device copy of a Byte-typed training image
				if (i < 0 || (size_t)i >= t.dataset.n_images) throw std::runtime_error{"Invalid frame index"};
				const NgpImageMeta& m = t.dataset.metadata[i];
				if (m.image_data_type != 1) throw std::runtime_error{"image is not RGBA8"};
				py::array_t<uint8_t> a({m.res[1], m.res[0], 4});
				t.dataset.pixelmemory[i].copy_to_host(a.mutable_data(), (size_t)m.res[0] * m.res[1] * 4);
				return a;
			}, py::arg("frame_idx"))
		.def("set_camera_intrinsics", &NerfTraining::set_camera_intrinsics, py::arg("frame_idx"), py::arg("fx") = 0.f, py::arg("fy") = 0.f, py::arg("cx") = -0.5f, py::arg("cy") = -0.5f,
			py::arg("k1") = 0.f, py::arg("k2") = 0.f, py::arg("p1") = 0.f, py::arg("p2") = 0.f);
}
