// ORACLE — TEST INFRASTRUCTURE ONLY.  AddressSanitizer / UndefinedBehaviourSanitizer run of the CPU restatement (GPU sanitizers are
// not available on the MI355X pool; the oracle shares the algorithms' index arithmetic with the kernels): a pile of mixed shapes
// with hinge, cone-twist and distance joints, 90 steps with each solver (scalar, 8-wide with its batch scheduler, custom order).
// Built and run by `make -C oracle sanitize` (tests/test_oracle.py::test_oracle_under_sanitizers).
#include "oworld.cpp"
#include <cstdio>

using namespace orc;

static unsigned long long g_state = 88172645463325252ull;
static float rnd(float lo, float hi) { g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17; return lo + (hi - lo) * (float)((g_state >> 11) & 0xFFFFFF) / (float)0x1000000; }

int main()
{
	const float mat[3] = { 0.1f, 0.5f, 1.f };
	for (u32 mode = 0; mode < 3; ++mode)
	{
		world* w = orc_world_create();
		const float ground[10] = { -30.f, -8.f, -30.f, 30.f, 0.f, 30.f }, origin[3] = { 0.f, 0.f, 0.f }, ident[4] = { 0.f, 0.f, 0.f, 1.f };
		orc_add_static_collider(w, 3, ground, mat, origin, ident);
		u32 prev = 0xFFFFFFFFu;
		for (u32 i = 0; i < 120; ++i)
		{
			float pos[3] = { 1.3f * (float)(i % 6) - 3.f + rnd(-0.05f, 0.05f), 0.8f + 1.3f * (float)(i / 24), 1.3f * (float)((i / 6) % 4) - 2.f + rnd(-0.05f, 0.05f) }; // a loose lattice: nothing overlaps at the start
			u32 b = orc_add_body(w, 0, 1.f, 0.4f, 0.4f, pos, ident);
			float shape[10] = { 0.f };
			u32 type = i % 4;
			if (type == 0) { shape[3] = rnd(0.2f, 0.5f); }                                                     // sphere
			else if (type == 1) { shape[0] = -0.3f; shape[3] = 0.3f; shape[6] = 0.2f; }                        // capsule
			else if (type == 2) { shape[1] = -0.3f; shape[4] = 0.3f; shape[6] = 0.25f; }                       // cylinder
			else { shape[3] = 1.f; shape[7] = rnd(0.2f, 0.5f); shape[8] = rnd(0.2f, 0.5f); shape[9] = rnd(0.2f, 0.5f); type = 4; } // obb
			orc_add_collider(w, b, type, shape, mat);
			if (prev != 0xFFFFFFFFu && i % 3 == 1)
			{
				const float axis[3] = { 0.f, 0.f, 1.f };
				if (i % 9 == 1) orc_add_hinge_constraint_global(w, prev, b, pos, axis, -0.5f, 0.5f);
				else if (i % 9 == 4) orc_add_cone_twist_constraint_global(w, prev, b, pos, axis, 0.6f, 0.4f);
				else { float other[3] = { pos[0] - 1.3f, pos[1], pos[2] }; orc_add_distance_constraint_global(w, prev, b, other, pos); }
			}
			prev = b;
		}
		std::vector<u32> order;
		for (u32 s = 0; s < 90; ++s)
		{
			if (mode == 2) // custom order: reversed emission order of the previous step's contact count (any permutation is valid input)
			{
				u32 n = orc_num_contacts(w);
				order.resize(n); for (u32 k = 0; k < n; ++k) order[k] = n - 1 - k;
				orc_set_custom_order(w, order.data(), n);
			}
			orc_step_internal(w, 30, mode, 1.f / 120.f);
		}
		std::vector<float> t(7 * (size_t)orc_num_bodies(w));
		orc_read_transforms(w, 1, t.data());
		for (float v : t) if (!(v == v)) { printf("NaN in mode %u\n", mode); return 1; }
		printf("mode %u: %u bodies, %u contacts, y of body 0 = %.4f\n", mode, orc_num_bodies(w), orc_num_contacts(w), t[1]);
		orc_world_destroy(w);
	}
	return 0;
}
