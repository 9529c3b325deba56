// CLI: render_amd [scene]  (reference: src/main.cpp)
#include <string>

#include "scene.h"

int main(int argc, char** argv)
{
	const std::string path = argc > 1 ? argv[1] : "scenes/cfg1_simple_shapes.scene";
	Scene(path).render();
	return 0;
}
