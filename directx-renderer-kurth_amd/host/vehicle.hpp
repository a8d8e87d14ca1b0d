// vehicle.hpp — the reference's gear-driven vehicle (src/physics/vehicle.h, vehicle.cpp:283-485) rebuilt over physics_facade.hpp:
// `vehicle::create(scene, initialMotorPosition, initialRotation)` with the same 16 named parts.  Physics only: the reference's
// mesh_builder / material calls are rendering and stay in the engine.  Gears collide through their capsule teeth (the disks are
// render-only), rods and wheel suspensions carry no colliders, the motor hinge is velocity-driven, the steering wheel hinge is
// position-driven (getConstraint(...).motorVelocity / .motorTargetAngle at run time, as in the reference).
// Same layout as directx-renderer-kurth_amd/scenes.py::add_vehicle, which the parity tests instantiate into the oracle too.
#pragma once
#include <cmath>
#include "physics_facade.hpp"

namespace mi
{
	namespace vehicle_math
	{
		inline vec3 operator+(vec3 a, vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
		inline vec3 operator-(vec3 a, vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
		inline vec3 operator*(vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
		inline vec3 cross(vec3 a, vec3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
		inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
		inline vec3 normalized(vec3 a) { return a * (1.f / std::sqrt(dot(a, a))); }
		inline quat axisAngle(vec3 axis, float angle) { float h = angle * 0.5f, s = std::sin(h); return { axis.x * s, axis.y * s, axis.z * s, std::cos(h) }; }
		inline quat operator*(quat a, quat b)
		{
			return { a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
				a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z };
		}
		inline vec3 operator*(quat q, vec3 v) { quat p{ v.x, v.y, v.z, 0.f }, c{ -q.x, -q.y, -q.z, q.w }; quat r = q * p * c; return { r.x, r.y, r.z }; }
		inline quat rotateFromTo(vec3 from, vec3 to) // shortest arc
		{
			from = normalized(from); to = normalized(to);
			float d = dot(from, to);
			if (d >= 1.f) return quat();
			if (d < 1e-6f - 1.f)
			{
				vec3 axis = cross(vec3(1.f, 0.f, 0.f), from);
				if (dot(axis, axis) == 0.f) axis = cross(vec3(0.f, 1.f, 0.f), from);
				return axisAngle(normalized(axis), 3.14159265358979f);
			}
			float s = std::sqrt((1.f + d) * 2.f);
			vec3 c = cross(from, to) * (1.f / s);
			quat q{ c.x, c.y, c.z, s * 0.5f };
			float l = 1.f / std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
			return { q.x * l, q.y * l, q.z * l, q.w * l };
		}
	}

	struct vehicle
	{
		scene_entity motor, motorGear, driveAxis, frontAxis, steeringWheel, steeringAxis, leftWheelSuspension, rightWheelSuspension,
			leftFrontWheel, rightFrontWheel, leftWheelArm, rightWheelArm, differentialSunGear, differentialSpiderGear, leftRearWheel, rightRearWheel;
		hinge_constraint_handle motorConstraint, steeringWheelConstraint; // the two joints the driver writes to
		scene_entity* parts() { return &motor; }
		static constexpr int numParts = 16;

		static vehicle create(game_scene& scene, vec3 initialMotorPosition, float initialRotation = 0.f) { vehicle v; v.initialize(scene, initialMotorPosition, initialRotation); return v; }

		void initialize(game_scene& scene, vec3 initialMotorPosition, float initialRotation = 0.f)
		{
			using namespace vehicle_math;
			const float density = 2000.f, deg = 3.14159265358979f / 180.f;
			const vec3 X(1.f, 0.f, 0.f), Y(0.f, 1.f, 0.f), Z(0.f, 0.f, 1.f);
			const quat place = axisAngle(Y, initialRotation);
			auto at = [&](vec3 p) { return place * p + initialMotorPosition; };
			auto dir = [&](vec3 d) { return place * d; };

			struct gear { float radius; uint32_t teeth; };               // height .1, tooth .07 x .1, friction 0: the same for every gear of the vehicle
			const float toothLength = 0.07f, toothWidth = 0.1f;
			const gear motorGearDesc{ 0.2f, 8 }, steeringWheelDesc{ 0.4f, 0 }, rearGearDesc{ 0.5f, 17 };
			const float wheelHeight = 0.3f, wheelRadius = 0.7f;

			auto body = [&](const char* name, vec3 position, quat rotation)
			{
				scene_entity e = scene.createEntity(name);
				e.addComponent<transform_component>(at(position), place * rotation);
				return e;
			};
			auto finish = [](scene_entity e) { e.addComponent<rigid_body_component>(false); return e; };
			auto teeth = [&](scene_entity e, gear g, float rodOffset)
			{
				for (uint32_t i = 0; i < g.teeth; ++i)
				{
					quat r = axisAngle(Y, (float)i * 6.28318530717959f / (float)g.teeth);
					vec3 center = r * vec3(g.radius + toothLength * 0.5f, 0.f, 0.f) + vec3(0.f, rodOffset, 0.f), half = r * vec3(toothLength * 0.5f, 0.f, 0.f);
					e.addComponent<collider_component>(collider_component::asCapsule(bounding_capsule{ center - half, center + half, toothWidth * 0.5f }, { 0.2f, 0.f, density }));
				}
			};
			auto wheel = [&](scene_entity e, float rodOffset)
			{
				e.addComponent<collider_component>(collider_component::asCylinder(bounding_cylinder{ vec3(0.f, rodOffset - wheelHeight * 0.5f, 0.f), vec3(0.f, rodOffset + wheelHeight * 0.5f, 0.f), wheelRadius }, { 0.2f, 1.f, 50.f }));
			};
			auto rod = [&](const char* name, vec3 from, vec3 to) { return finish(body(name, (from + to) * 0.5f, rotateFromTo(Y, to - from))); };

			const float motorGearY = 0.25f, gearOffset = 0.26f;

			motor = body("Motor", vec3(), quat());
			motor.addComponent<collider_component>(collider_component::asAABB(bounding_box::fromCenterRadius(vec3(), vec3(0.6f, 0.1f, 1.f)), { 0.2f, 0.f, density }));
			finish(motor);

			motorGear = body("Axis", vec3(0.f, motorGearY, 0.f), quat());
			teeth(motorGear, motorGearDesc, 0.f); finish(motorGear);
			motorConstraint = addHingeConstraintFromGlobalPoints(motor, motorGear, at(vec3(0.f, motorGearY, 0.f)), dir(Y));
			{ auto c = getConstraint(scene, motorConstraint); c->maxMotorTorque = 500.f; c->motorVelocity = 0.f; }

			const float driveAxisLength = 4.5f;
			driveAxis = body("Axis", vec3(0.f, motorGearY + gearOffset, gearOffset), axisAngle(vec3(-1.f, 0.f, 0.f), 90.f * deg));
			teeth(driveAxis, motorGearDesc, 0.f); teeth(driveAxis, motorGearDesc, -(driveAxisLength * 0.57f - 1.1f)); finish(driveAxis);
			addHingeConstraintFromGlobalPoints(motor, driveAxis, at(vec3(0.f, motorGearY + gearOffset, gearOffset)), dir(Z));

			const float axisLength = 1.5f, suspensionLength = 0.4f;
			const float frontAxisOffsetZ = -driveAxisLength * 0.5f + gearOffset * 2.f;
			const vec3 frontAxisPos(0.f, motorGearY + gearOffset, frontAxisOffsetZ);
			frontAxis = rod("Rod", frontAxisPos + vec3(axisLength, 0.f, 0.f), frontAxisPos - vec3(axisLength, 0.f, 0.f));
			addFixedConstraintFromGlobalPoints(motor, frontAxis, at(frontAxisPos));

			const quat steeringWheelRot = axisAngle(vec3(-1.f, 0.f, 0.f), -80.f * deg);
			const vec3 steeringWheelPos(0.f, 1.12f, 0.81f);
			steeringWheel = body("Axis", steeringWheelPos, steeringWheelRot);
			teeth(steeringWheel, steeringWheelDesc, 0.f); teeth(steeringWheel, motorGearDesc, -2.f); finish(steeringWheel);
			steeringWheelConstraint = addHingeConstraintFromGlobalPoints(motor, steeringWheel, at(steeringWheelPos), dir(steeringWheelRot * vec3(0.f, -1.f, 0.f)));
			{ auto c = getConstraint(scene, steeringWheelConstraint); c->motorType = constraint_position_motor; c->maxMotorTorque = 1000.f; c->motorTargetAngle = 0.f; }

			const vec3 steeringAxisPos(0.f, motorGearY + gearOffset + 0.06f, frontAxisOffsetZ + 0.49f);
			const float steeringAxisLength = axisLength * 1.05f;
			steeringAxis = body("Gear Axis", steeringAxisPos, steeringWheelRot);
			for (uint32_t i = 0; i < 8; ++i) // the rack
			{
				float x = -0.5f * steeringAxisLength + 0.5f * toothWidth + (float)i * ((steeringAxisLength - toothWidth) / 7.f);
				steeringAxis.addComponent<collider_component>(collider_component::asCapsule(bounding_capsule{ vec3(x, toothWidth * 0.5f + toothLength * 0.5f, 0.f), vec3(x, toothWidth * 0.5f - toothLength * 0.5f, 0.f), toothWidth * 0.5f }, { 0.2f, 0.f, density }));
			}
			finish(steeringAxis);
			addSliderConstraintFromGlobalPoints(motor, steeringAxis, at(steeringAxisPos), dir(X), -4.f, 4.f);
			const vec3 leftRackEnd = steeringAxisPos - vec3(steeringAxisLength * 0.5f, 0.f, 0.f), rightRackEnd = steeringAxisPos + vec3(steeringAxisLength * 0.5f, 0.f, 0.f);

			const vec3 leftSuspensionPos = frontAxisPos - vec3(axisLength, 0.f, 0.f), leftSuspensionAttachment = leftSuspensionPos + vec3(0.f, 0.f, suspensionLength);
			leftWheelSuspension = finish(body("Wheel suspension", leftSuspensionPos, quat()));
			addHingeConstraintFromGlobalPoints(motor, leftWheelSuspension, at(leftSuspensionPos), dir(Y), -45.f * deg, 45.f * deg);
			const vec3 rightSuspensionPos = frontAxisPos + vec3(axisLength, 0.f, 0.f), rightSuspensionAttachment = rightSuspensionPos + vec3(0.f, 0.f, suspensionLength);
			rightWheelSuspension = finish(body("Wheel suspension", rightSuspensionPos, quat()));
			addHingeConstraintFromGlobalPoints(motor, rightWheelSuspension, at(rightSuspensionPos), dir(Y), -45.f * deg, 45.f * deg);

			const quat wheelRot = axisAngle(Z, 90.f * deg);
			const vec3 leftFrontWheelPos = leftSuspensionPos - vec3(suspensionLength * 0.5f, 0.f, 0.f), rightFrontWheelPos = rightSuspensionPos + vec3(suspensionLength * 0.5f, 0.f, 0.f);
			leftFrontWheel = body("Wheel", leftFrontWheelPos, wheelRot); wheel(leftFrontWheel, 0.f); finish(leftFrontWheel);
			rightFrontWheel = body("Wheel", rightFrontWheelPos, wheelRot); wheel(rightFrontWheel, 0.f); finish(rightFrontWheel);
			addHingeConstraintFromGlobalPoints(leftFrontWheel, leftWheelSuspension, at(leftFrontWheelPos), dir(X));
			addHingeConstraintFromGlobalPoints(rightFrontWheel, rightWheelSuspension, at(rightFrontWheelPos), dir(X));

			leftWheelArm = rod("Rod", leftRackEnd, leftSuspensionAttachment);
			rightWheelArm = rod("Rod", rightRackEnd, rightSuspensionAttachment);
			addBallConstraintFromGlobalPoints(leftWheelSuspension, leftWheelArm, at(leftSuspensionAttachment));
			addBallConstraintFromGlobalPoints(steeringAxis, leftWheelArm, at(leftRackEnd));
			addBallConstraintFromGlobalPoints(rightWheelSuspension, rightWheelArm, at(rightSuspensionAttachment));
			addBallConstraintFromGlobalPoints(steeringAxis, rightWheelArm, at(rightRackEnd));

			const float rearZ = driveAxisLength * 0.505f;
			const vec3 sunPos(-gearOffset, motorGearY + gearOffset, rearZ);
			differentialSunGear = body("Axis", sunPos, axisAngle(vec3(0.f, 0.f, -1.f), 90.f * deg));
			teeth(differentialSunGear, rearGearDesc, 0.f); finish(differentialSunGear);
			addHingeConstraintFromGlobalPoints(motor, differentialSunGear, at(sunPos), dir(X));

			const vec3 spiderPos(0.11f, motorGearY + gearOffset * 2.f, rearZ);
			differentialSpiderGear = body("Axis", spiderPos, quat());
			teeth(differentialSpiderGear, motorGearDesc, 0.f); finish(differentialSpiderGear);
			addHingeConstraintFromGlobalPoints(differentialSunGear, differentialSpiderGear, at(spiderPos), dir(Y));

			const vec3 leftRearWheelPos = spiderPos + vec3(-gearOffset, -gearOffset, 0.f), rightRearWheelPos = spiderPos + vec3(gearOffset, -gearOffset, 0.f);
			const quat rearRot = axisAngle(vec3(0.f, 0.f, -1.f), 90.f * deg);
			leftRearWheel = body("Axis", leftRearWheelPos, rearRot);
			teeth(leftRearWheel, motorGearDesc, 0.f); wheel(leftRearWheel, -(axisLength + spiderPos.x)); finish(leftRearWheel);
			rightRearWheel = body("Axis", rightRearWheelPos, rearRot);
			teeth(rightRearWheel, motorGearDesc, 0.f); wheel(rightRearWheel, axisLength - spiderPos.x); finish(rightRearWheel);
			addHingeConstraintFromGlobalPoints(motor, leftRearWheel, at(leftRearWheelPos), dir(X));
			addHingeConstraintFromGlobalPoints(motor, rightRearWheel, at(rightRearWheelPos), dir(X));
		}
	};
}
