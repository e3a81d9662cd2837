// cam_adam.h — the small host-side Adam optimizers of the per-camera trainables (include/neural-graphics-primitives/adam_optimizer.h:23-162):
// AdamOptimizer<Vector3f> for the position offsets and RotationAdamOptimizer for the angle-axis rotation offsets of optimize_extrinsics.
// Scalar fp32 throughout, like the reference (a handful of floats per image every n_steps_between_cam_updates steps: host work).
//
// The reference composes rotations with Eigen (dependencies/eigen, an un-vendored submodule; Eigen 3.4 restated here):
//   AngleAxis::toRotationMatrix  (Eigen/src/Geometry/AngleAxis.h)   — Rodrigues, element by element as written there
//   Quaternion(Matrix3)          (Eigen/src/Geometry/Quaternion.h)  — trace > 0 branch, else largest diagonal element
//   AngleAxis(Quaternion)        (Eigen/src/Geometry/AngleAxis.h)   — angle = 2 atan2(|vec|, |w|), axis = vec / (+-|vec|); zero rotation -> axis (1,0,0)
#pragma once
#include <cmath>
#include <cstdint>

namespace ngp {

// 3x3 column-major: element (r, c) = m[3 * c + r] — the layout of the rotation block of a 3x4 camera matrix
inline void angle_axis_to_matrix(float angle, const float axis[3], float m[9]) {
	const float s = std::sin(angle), c = std::cos(angle);
	const float sin_axis[3] = {s * axis[0], s * axis[1], s * axis[2]};
	const float cos1_axis[3] = {(1.0f - c) * axis[0], (1.0f - c) * axis[1], (1.0f - c) * axis[2]};
	float tmp;
	tmp = cos1_axis[0] * axis[1]; m[3 * 1 + 0] = tmp - sin_axis[2]; m[3 * 0 + 1] = tmp + sin_axis[2];
	tmp = cos1_axis[0] * axis[2]; m[3 * 2 + 0] = tmp + sin_axis[1]; m[3 * 0 + 2] = tmp - sin_axis[1];
	tmp = cos1_axis[1] * axis[2]; m[3 * 2 + 1] = tmp - sin_axis[0]; m[3 * 1 + 2] = tmp + sin_axis[0];
	m[0] = cos1_axis[0] * axis[0] + c; m[4] = cos1_axis[1] * axis[1] + c; m[8] = cos1_axis[2] * axis[2] + c;
}
inline void mat3_mul(const float a[9], const float b[9], float out[9]) {   // out = a * b (column-major), rows summed left to right
	float r[9];
	for (int c = 0; c < 3; ++c) for (int row = 0; row < 3; ++row) r[3 * c + row] = a[row] * b[3 * c] + a[3 + row] * b[3 * c + 1] + a[6 + row] * b[3 * c + 2];
	for (int i = 0; i < 9; ++i) out[i] = r[i];
}
inline void matrix_to_angle_axis(const float m[9], float& angle, float axis[3]) {
	auto at = [&](int r, int c) { return m[3 * c + r]; };
	float q[4];   // x y z w
	float t = at(0, 0) + at(1, 1) + at(2, 2);
	if (t > 0.0f) {
		t = std::sqrt(t + 1.0f);
		q[3] = 0.5f * t;
		t = 0.5f / t;
		q[0] = (at(2, 1) - at(1, 2)) * t; q[1] = (at(0, 2) - at(2, 0)) * t; q[2] = (at(1, 0) - at(0, 1)) * t;
	} else {
		int i = 0;
		if (at(1, 1) > at(0, 0)) i = 1;
		if (at(2, 2) > at(i, i)) i = 2;
		const int j = (i + 1) % 3, k = (j + 1) % 3;
		t = std::sqrt(at(i, i) - at(j, j) - at(k, k) + 1.0f);
		q[i] = 0.5f * t;
		t = 0.5f / t;
		q[3] = (at(k, j) - at(j, k)) * t; q[j] = (at(j, i) + at(i, j)) * t; q[k] = (at(k, i) + at(i, k)) * t;
	}
	float n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
	if (n != 0.0f) {
		angle = 2.0f * std::atan2(n, std::fabs(q[3]));
		if (q[3] < 0.0f) n = -n;
		axis[0] = q[0] / n; axis[1] = q[1] / n; axis[2] = q[2] / n;
	} else {
		angle = 0.0f; axis[0] = 1.0f; axis[1] = 0.0f; axis[2] = 0.0f;
	}
}

struct AdamHyper { float learning_rate, epsilon = 1e-8f, beta1 = 0.9f, beta2 = 0.99f; };

// AdamOptimizer<Vector3f> (adam_optimizer.h:23-101)
struct Vec3Adam {
	uint32_t iter = 0;
	float first_moment[3] = {0, 0, 0}, second_moment[3] = {0, 0, 0}, variable[3] = {0, 0, 0};
	AdamHyper h{1e-4f};
	void reset_state() { iter = 0; for (int c = 0; c < 3; ++c) first_moment[c] = second_moment[c] = variable[c] = 0.f; }
	void step(const float gradient[3]) {
		++iter;
		const float actual_learning_rate = h.learning_rate * std::sqrt(1 - std::pow(h.beta2, (float)iter)) / (1 - std::pow(h.beta1, (float)iter));
		for (int c = 0; c < 3; ++c) {
			first_moment[c] = h.beta1 * first_moment[c] + (1 - h.beta1) * gradient[c];
			second_moment[c] = h.beta2 * second_moment[c] + (1 - h.beta2) * (gradient[c] * gradient[c]);
			variable[c] -= actual_learning_rate * (first_moment[c] / (std::sqrt(second_moment[c]) + h.epsilon));
		}
	}
};

// RotationAdamOptimizer (adam_optimizer.h:103-162): Adam's update is itself an angle-axis rotation that is composed with the variable
struct RotationAdam {
	uint32_t iter = 0;
	float first_moment[3] = {0, 0, 0}, second_moment[3] = {0, 0, 0}, variable[3] = {0, 0, 0};
	AdamHyper h{1e-4f};
	void reset_state() { iter = 0; for (int c = 0; c < 3; ++c) first_moment[c] = second_moment[c] = variable[c] = 0.f; }
	void step(const float gradient[3]) {
		++iter;
		// std::pow(float, uint32_t) promotes to double (adam_optimizer.h:121): the bias correction is evaluated in double and rounded once
		const float actual_learning_rate = (float)((double)h.learning_rate * std::sqrt(1 - std::pow((double)h.beta2, (double)iter)) / (1 - std::pow((double)h.beta1, (double)iter)));
		float rot[3];
		for (int c = 0; c < 3; ++c) {
			first_moment[c] = h.beta1 * first_moment[c] + (1 - h.beta1) * gradient[c];
			second_moment[c] = h.beta2 * second_moment[c] + (1 - h.beta2) * (gradient[c] * gradient[c]);
			rot[c] = actual_learning_rate * (first_moment[c] / (std::sqrt(second_moment[c]) + h.epsilon));
		}
		const float rot_len = std::sqrt(rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2]);
		const float var_len = std::sqrt(variable[0] * variable[0] + variable[1] * variable[1] + variable[2] * variable[2]);
		const float Z[3] = {0.f, 0.f, 1.f};
		float a_rot[3], a_var[3];
		for (int c = 0; c < 3; ++c) { a_rot[c] = rot_len > 0 ? rot[c] / rot_len : Z[c]; a_var[c] = var_len > 0 ? variable[c] / var_len : Z[c]; }
		float m_rot[9], m_var[9], m[9];
		angle_axis_to_matrix(-rot_len, a_rot, m_rot);
		angle_axis_to_matrix(var_len, a_var, m_var);
		mat3_mul(m_rot, m_var, m);
		float angle, axis[3];
		matrix_to_angle_axis(m, angle, axis);
		for (int c = 0; c < 3; ++c) variable[c] = axis[c] * angle;
	}
};

} // namespace ngp
