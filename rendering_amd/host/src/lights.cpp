// Area-light sample grid (reference: src/lights.cpp:46-63); the other light types carry parameters only.
#include "lights.h"

void AreaLight::setPoints()
{
	if (pointsCreated) return;
	pointsCreated = true;
	const Vec3f corner = pos - (i / 2.0f) - (j / 2.0f);
	if (samples > 1) {
		for (int a = 0; a < samples; ++a)
			for (int b = 0; b < samples; ++b)
				points.push_back(corner + (i * (((float)a) / (samples - 1))) + (j * (((float)b) / (samples - 1))));
	}
	else points.push_back(pos);
}
