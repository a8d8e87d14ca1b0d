// example_vehicle.cpp — vehicle::create (reference vehicle.h) over physics_facade.hpp: builds the gear-driven vehicle on a platform, runs
// the motor, prints every part's mass properties and pose as built and its pose after 360 steps (tests/test_gpu_vehicle.py compares them
// with the same vehicle built by scenes.add_vehicle through the ctypes mirror).
#include <cstdio>
#include "vehicle.hpp"

using namespace mi;

static void printParts(const char* tag, vehicle& v)
{
	for (int i = 0; i < vehicle::numParts; ++i)
	{
		auto t = v.parts()[i].transform();
		std::printf("%s %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n", tag, t.position.x, t.position.y, t.position.z, t.rotation.x, t.rotation.y, t.rotation.z, t.rotation.w);
	}
}

int main()
{
	try
	{
		game_scene scene;
		scene.createEntity("platform").addComponent<collider_component>(collider_component::asAABB(bounding_box{ vec3(-40.f, -8.f, -40.f), vec3(40.f, 0.f, 40.f) }, { 0.1f, 1.f, 1.f }));
		vehicle v = vehicle::create(scene, vec3(0.f, 1.1f, 0.f), 0.3f);
		getConstraint(scene, v.motorConstraint)->motorVelocity = 3.f;

		memory_arena arena; physics_settings settings; float timer = 0.f;
		printParts("init", v);
		{
			float mass[16 * 13];
			scene.check(mi_read_mass_properties(scene.world, mass, 16), "mi_read_mass_properties");
			for (int i = 0; i < 16; ++i) { std::printf("mass"); for (int k = 0; k < 13; ++k) std::printf(" %.9g", mass[13 * i + k]); std::printf("\n"); }
		}
		for (int frame = 0; frame < 360; ++frame) physicsStep(scene, arena, timer, settings, 1.f / 120.f);
		printParts("late", v);
		return 0;
	}
	catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
}
