// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's cloth (src/physics/cloth.h, cloth.cpp:7-341): a grid of
// particles with stretch / shear / bend distance constraints, stepped after the rigid bodies (physics.cpp:1354-1358).
// Two Gauss-Seidel orders: the reference's storage order, and the colour order the device runs (12 colours: constraint family x one
// parity bit; inside a colour no two constraints share a particle, so solving a colour in parallel equals solving it in sequence).
#pragma once
#include "omath.h"
#include <vector>
#include <algorithm>

namespace orc
{

struct cloth_constraint { u32 a, b; float restDistance, inverseMassSum; };
struct cloth_constraint_temp { vec3 gradient; float inverseScaledGradientSquared; };

struct cloth
{
	float totalMass, gravityFactor, damping, stiffness;
	u32 gridSizeX, gridSizeY;
	float width, height;
	float oldTotalMass, oldStiffness;
	std::vector<vec3> positions, prevPositions, velocities, forceAccumulators;
	std::vector<float> invMasses;
	std::vector<cloth_constraint> constraints;
	std::vector<u32> colour;       // per constraint: family * 2 + parity bit
	std::vector<u32> colourOrder;  // constraint indices sorted by colour (stable)

	vec3 getParticlePosition(float relX, float relY) const // cloth.cpp:134-140
	{
		vec3 position = vec3(relX * width, -relY * height, 0.f);
		position.x -= width * 0.5f;
		std::swap(position.y, position.z);
		return position;
	}
	void addConstraint(u32 indexA, u32 indexB, u32 col) // cloth.cpp:320-329
	{
		constraints.push_back(cloth_constraint{ indexA, indexB, length(positions[indexA] - positions[indexB]), (invMasses[indexA] + invMasses[indexB]) / stiffness });
		colour.push_back(col);
	}
	cloth(float width_, float height_, u32 gx, u32 gy, float totalMass_, float stiffness_, float damping_, float gravityFactor_) // cloth.cpp:7-88
		: totalMass(totalMass_), gravityFactor(gravityFactor_), damping(damping_), stiffness(stiffness_), gridSizeX(gx), gridSizeY(gy), width(width_), height(height_)
	{
		u32 numParticles = gridSizeX * gridSizeY;
		float invMassPerParticle = numParticles / totalMass;
		velocities.resize(numParticles, vec3(0.f));
		forceAccumulators.resize(numParticles, vec3(0.f));
		for (u32 y = 0; y < gridSizeY; ++y)
		{
			float invMass = (y == 0) ? 0.f : invMassPerParticle; // the upper row is locked
			for (u32 x = 0; x < gridSizeX; ++x)
			{
				float relX = x / (float)(gridSizeX - 1);
				float relY = y / (float)(gridSizeY - 1);
				vec3 position = getParticlePosition(relX, relY);
				positions.push_back(position); prevPositions.push_back(position); invMasses.push_back(invMass);
			}
		}
		for (u32 y = 0; y < gridSizeY; ++y)
		{
			for (u32 x = 0; x < gridSizeX; ++x)
			{
				u32 index = y * gridSizeX + x;
				if (x < gridSizeX - 1) { addConstraint(index, index + 1, 0 + (x & 1)); }                          // stretch
				if (y < gridSizeY - 1) { addConstraint(index, index + gridSizeX, 2 + (y & 1)); }
				if (x < gridSizeX - 1 && y < gridSizeY - 1)                                                      // shear
				{
					addConstraint(index, index + gridSizeX + 1, 4 + (x & 1));
					addConstraint(index + gridSizeX, index + 1, 6 + (x & 1));
				}
				if (x + 2 < gridSizeX) { addConstraint(index, index + 2, 8 + ((x >> 1) & 1)); }                  // bend (reference: x < gridSizeX - 2)
				if (y + 2 < gridSizeY) { addConstraint(index, index + gridSizeX * 2, 10 + ((y >> 1) & 1)); }
			}
		}
		colourOrder.resize(constraints.size());
		for (u32 i = 0; i < colourOrder.size(); ++i) colourOrder[i] = i;
		std::stable_sort(colourOrder.begin(), colourOrder.end(), [this](u32 l, u32 r) { return colour[l] < colour[r]; });
		oldTotalMass = totalMass; oldStiffness = stiffness;
	}

	void setWorldPositionOfFixedVertices(const trs& transform, bool moveRigid) // cloth.cpp:90-132
	{
		if (moveRigid)
		{
			vec3 pivot;
			if (gridSizeX % 2 == 1) { pivot = positions[gridSizeX / 2]; }
			else { pivot = (positions[gridSizeX / 2] + positions[gridSizeX / 2 - 1]) * 0.5f; }
			vec3 currentAxis = normalize(positions[gridSizeX - 1] - positions[0]);
			vec3 newAxis = normalize(transformPosition(transform, getParticlePosition(1.f, 0.f)) - transformPosition(transform, getParticlePosition(0.f, 0.f)));
			vec3 newPivot = transformPosition(transform, getParticlePosition(0.5f, 0.f));
			quat deltaRotation = rotateFromTo(currentAxis, newAxis);
			for (u32 y = 1; y < gridSizeY; ++y)
				for (u32 x = 0; x < gridSizeX; ++x)
				{
					vec3& position = positions[y * gridSizeX + x];
					position = deltaRotation * (position - pivot) + newPivot;
				}
		}
		for (u32 x = 0; x < gridSizeX; ++x)
		{
			float relX = x / (float)(gridSizeX - 1);
			positions[x] = transformPosition(transform, getParticlePosition(relX, 0.f));
		}
	}

	void applyWindForce(vec3 force) // cloth.cpp:147-186
	{
		for (u32 y = 0; y < gridSizeY - 1; ++y)
		{
			for (u32 x = 0; x < gridSizeX - 1; ++x)
			{
				u32 tl = y * gridSizeX + x, tr = tl + 1, bl = tl + gridSizeX, br = bl + 1;
				{
					vec3 normal = cross(positions[bl] - positions[tl], positions[tr] - positions[tl]);
					vec3 forceInNormalDir = normal * dot(normalize(normal), force);
					forceInNormalDir *= 1.f / 3.f;
					forceAccumulators[tl] += forceInNormalDir; forceAccumulators[tr] += forceInNormalDir; forceAccumulators[bl] += forceInNormalDir;
				}
				{
					vec3 normal = cross(positions[tr] - positions[br], positions[bl] - positions[br]);
					vec3 forceInNormalDir = normal * dot(normalize(normal), force);
					forceInNormalDir *= 1.f / 3.f;
					forceAccumulators[br] += forceInNormalDir; forceAccumulators[tr] += forceInNormalDir; forceAccumulators[bl] += forceInNormalDir;
				}
			}
		}
	}

	void recalculateProperties() // cloth.cpp:331-347
	{
		u32 numParticles = gridSizeX * gridSizeY;
		float invMassPerParticle = numParticles / totalMass;
		for (float& invMass : invMasses) { invMass = (invMass != 0.f) ? invMassPerParticle : 0.f; }
		stiffness = std::min(std::max(stiffness, 0.01f), 1.f);
		float invStiffness = 1.f / stiffness;
		for (cloth_constraint& c : constraints) { c.inverseMassSum = (invMasses[c.a] + invMasses[c.b]) * invStiffness; }
	}

	void solveVelocity(u32 i, const std::vector<cloth_constraint_temp>& temp) // cloth.cpp:289-299
	{
		cloth_constraint& c = constraints[i];
		float j = -dot(temp[i].gradient, velocities[c.a] - velocities[c.b]) * temp[i].inverseScaledGradientSquared;
		velocities[c.a] += temp[i].gradient * (j * invMasses[c.a]);
		velocities[c.b] -= temp[i].gradient * (j * invMasses[c.b]);
	}
	void solvePosition(u32 i) // cloth.cpp:301-318
	{
		cloth_constraint& c = constraints[i];
		if (c.inverseMassSum > 0.f)
		{
			vec3 delta = positions[c.b] - positions[c.a];
			float len = squaredLength(delta);
			float sqRestDistance = c.restDistance * c.restDistance;
			if (sqRestDistance + len > 1e-5f)
			{
				float k = ((sqRestDistance - len) / (c.inverseMassSum * (sqRestDistance + len)));
				positions[c.a] -= delta * (k * invMasses[c.a]);
				positions[c.b] += delta * (k * invMasses[c.b]);
			}
		}
	}

	// cloth.cpp:194-287.  colourOrdered = false: the reference's storage order; true: the device's colour order.
	void simulate(u32 velocityIterations, u32 positionIterations, u32 driftIterations, float dt, bool colourOrdered)
	{
		if (totalMass != oldTotalMass || stiffness != oldStiffness) { recalculateProperties(); oldTotalMass = totalMass; oldStiffness = stiffness; }
		const float GRAVITY = -9.81f;
		float gravityVelocity = GRAVITY * dt * gravityFactor;
		u32 numParticles = gridSizeX * gridSizeY, numConstraints = (u32)constraints.size();
		auto at = [&](u32 k) { return colourOrdered ? colourOrder[k] : k; };
		for (u32 i = 0; i < numParticles; ++i)
		{
			if (invMasses[i] > 0.f) { velocities[i].y += gravityVelocity; }
			velocities[i] += forceAccumulators[i] * (invMasses[i] * dt);
			prevPositions[i] = positions[i];
			positions[i] += velocities[i] * dt;
			forceAccumulators[i] = vec3(0.f);
		}
		float invDt = (dt > 1e-5f) ? (1.f / dt) : 1.f;
		if (velocityIterations > 0)
		{
			std::vector<cloth_constraint_temp> temp(numConstraints);
			for (u32 i = 0; i < numConstraints; ++i)
			{
				const cloth_constraint& c = constraints[i];
				temp[i].gradient = prevPositions[c.b] - prevPositions[c.a];
				temp[i].inverseScaledGradientSquared = (c.inverseMassSum == 0.f) ? 0.f : (1.f / (squaredLength(temp[i].gradient) * c.inverseMassSum));
			}
			for (u32 it = 0; it < velocityIterations; ++it) for (u32 k = 0; k < numConstraints; ++k) solveVelocity(at(k), temp);
			for (u32 i = 0; i < numParticles; ++i) { positions[i] = prevPositions[i] + velocities[i] * dt; }
		}
		if (positionIterations > 0)
		{
			for (u32 it = 0; it < positionIterations; ++it) for (u32 k = 0; k < numConstraints; ++k) solvePosition(at(k));
			for (u32 i = 0; i < numParticles; ++i) { velocities[i] = (positions[i] - prevPositions[i]) * invDt; }
		}
		if (driftIterations > 0)
		{
			for (u32 i = 0; i < numParticles; ++i) { prevPositions[i] = positions[i]; }
			for (u32 it = 0; it < driftIterations; ++it) for (u32 k = 0; k < numConstraints; ++k) solvePosition(at(k));
			for (u32 i = 0; i < numParticles; ++i) { velocities[i] += (positions[i] - prevPositions[i]) * invDt; }
		}
		float dampingFactor = 1.f / (1.f + dt * damping);
		for (u32 i = 0; i < numParticles; ++i) { velocities[i] *= dampingFactor; }
	}
};

} // namespace orc
