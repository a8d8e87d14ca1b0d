// example_facade.cpp — the reference's call shapes (physics.h / rigid_body.h / scene.h) driving the HIP world through physics_facade.hpp.
// Builds a ground platform (application.cpp:209-212), a small box pile and a hinged pendulum, steps 120 frames of 1/60 s through
// physicsStep() and prints the final poses (tests/test_gpu_facade.py compares them with the same world built through the ctypes mirror).
// Build: g++ -std=c++17 -Iinclude example_facade.cpp -L.. -lmi_physics
#include <cstdio>
#include <cstdlib>
#include "physics_facade.hpp"

using namespace mi;

int main()
{
	try
	{
		game_scene scene;
		physics_material mat{ 0.1f, 0.5f, 1.f };
		scene.createEntity("platform")
			.addComponent<transform_component>(vec3(0.f, -4.f, 0.f), quat())
			.addComponent<collider_component>(collider_component::asAABB(bounding_box::fromCenterRadius(vec3(0.f, 0.f, 0.f), vec3(30.f, 4.f, 30.f)), mat));

		std::vector<scene_entity> boxes;
		for (int i = 0; i < 8; ++i)
		{
			auto e = scene.createEntity("box");
			e.addComponent<transform_component>(vec3(0.1f * i, 1.f + 2.5f * i, 0.05f * i), quat())
				.addComponent<collider_component>(collider_component::asOBB(bounding_oriented_box{ quat(), vec3(0.f, 0.f, 0.f), vec3(1.f, 0.5f, 0.75f) }, mat))
				.addComponent<rigid_body_component>(false, 1.f);
			boxes.push_back(e);
		}
		auto anchor = scene.createEntity("anchor");
		anchor.addComponent<transform_component>(vec3(10.f, 6.f, 0.f), quat())
			.addComponent<collider_component>(collider_component::asSphere(bounding_sphere{ vec3(0.f, 0.f, 0.f), 0.25f }, mat))
			.addComponent<rigid_body_component>(true, 1.f);
		auto bob = scene.createEntity("bob");
		bob.addComponent<transform_component>(vec3(12.f, 6.f, 0.f), quat())
			.addComponent<collider_component>(collider_component::asCapsule(bounding_capsule{ vec3(-0.5f, 0.f, 0.f), vec3(0.5f, 0.f, 0.f), 0.3f }, mat))
			.addComponent<rigid_body_component>(false, 1.f);
		auto hinge = addHingeConstraintFromGlobalPoints(anchor, bob, vec3(10.f, 6.f, 0.f), vec3(0.f, 0.f, 1.f));
		{
			auto h = getConstraint(scene, hinge); // write-back proxy: same field writes as the reference's T&
			h->motorType = constraint_velocity_motor; h->motorVelocity = 0.5f; h->maxMotorTorque = 50.f;
		}

		// addConstraint(a, b, const T&) (physics.h:239-244), as the deserialisers call it: a second bob held by a ready-made distance constraint
		auto bob2 = scene.createEntity("bob2");
		bob2.addComponent<transform_component>(vec3(10.f, 4.f, 0.f), quat())
			.addComponent<collider_component>(collider_component::asSphere(bounding_sphere{ vec3(0.f, 0.f, 0.f), 0.3f }, mat))
			.addComponent<rigid_body_component>(false, 1.f);
		distance_constraint rope{ vec3(0.f, 0.f, 0.f), vec3(0.f, 0.f, 0.f), 2.f };
		auto ropeHandle = addConstraint(anchor, bob2, rope);
		if (getConstraint(scene, ropeHandle)->globalLength != 2.f) std::abort();

		// a global wind, an updraft box over the pile, and a trigger slab the boxes fall through (physics.h:182-203)
		scene.createEntity("wind").addComponent<force_field_component>(vec3(0.5f, 0.f, 0.f));
		scene.createEntity("updraft")
			.addComponent<transform_component>(vec3(0.f, 6.f, 0.f), quat())
			.addComponent<collider_component>(collider_component::asAABB(bounding_box::fromCenterRadius(vec3(0.f, 0.f, 0.f), vec3(2.f, 1.f, 2.f)), mat))
			.addComponent<force_field_component>(vec3(0.f, 8.f, 0.f));
		int enters = 0, leaves = 0, begins = 0, ends = 0;
		auto slab = scene.createEntity("slab");
		slab.addComponent<transform_component>(vec3(0.f, 3.f, 0.f), quat())
			.addComponent<trigger_component>([&](trigger_event e) { (e.type == trigger_event_enter ? enters : leaves)++; if (!(e.trigger == slab)) std::abort(); })
			.addComponent<collider_component>(collider_component::asAABB(bounding_box::fromCenterRadius(vec3(0.f, 0.f, 0.f), vec3(3.f, 0.25f, 3.f)), mat));

		{	// a flat terrain chunk well below the platform: exercises the heightmap entry points
			scene.createEntity("terrain").addComponent<heightmap_collider_component>(1u, 64.f, mat);
			std::vector<uint16_t> heights(129 * 129, 1000);
			scene.heightmapSetHeights(0, 0, heights.data());
			scene.heightmapUpdate(vec3(-32.f, -20.f, -32.f), 2.f);
			if (!(scene.heightmapHeightAt(0.f, 0.f) < -19.9f)) std::abort();
		}
		auto banner = scene.createEntity("banner");
		banner.addComponent<transform_component>(vec3(-8.f, 9.f, 0.f), quat()).addComponent<cloth_component>(4.f, 3.f, 12u, 9u, 2.f);

		memory_arena arena; physics_settings settings; float timer = 0.f;
		settings.collisionBeginCallback = [&](const collision_begin_event& e) { ++begins; if (e.colliderA.type > e.colliderB.type) std::abort(); };
		settings.collisionEndCallback = [&](const collision_end_event&) { ++ends; };
		for (int frame = 0; frame < 120; ++frame) physicsStep(scene, arena, timer, settings, 1.f / 60.f);

		for (auto& e : boxes) { auto t = e.transform(); std::printf("box %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n", t.position.x, t.position.y, t.position.z, t.rotation.x, t.rotation.y, t.rotation.z, t.rotation.w); }
		std::printf("events %d %d %d %d\n", enters, leaves, begins, ends);
		{ auto p = banner.clothPositions(); vec3 c = p.back(); std::printf("cloth %.6f %.6f %.6f\n", c.x, c.y, c.z); }
		auto t = bob.transform(); std::printf("bob %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n", t.position.x, t.position.y, t.position.z, t.rotation.x, t.rotation.y, t.rotation.z, t.rotation.w);
		return 0;
	}
	catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
}
