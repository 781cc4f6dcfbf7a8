// rt_main.cpp -- what the reference's main() (main.cpp:538-662) becomes on top of the GPU renderer: the scene
// functions fill the globals `objects`, `lights`, `ambient_light` exactly as before, the tile loop is ONE call to
// render(), and the compile-time comment toggles of the reference (resolution :543-544, scene :552-555, SSAA count
// :625, bounce depth :643) are run-time arguments:
//
//   raytracer_hip [--scene sample|bunny|competition|cones] [--width W] [--height H] [--spp N] [--depth D]
//                 [--assets DIR] [--p3] [--gpus N] [output.ppm]
//
// Output: binary P6 by default (an 8K frame as ASCII P3 is ~400 MB), the reference's ASCII P3 with --p3.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "Scene.h"

int main(int argc, const char *argv[]) {
    std::string scene = "competition", assets = ".", out = "./result.ppm";
    int width = 1024, height = 768, spp = 4, depth = 3;  // main.cpp:543-545,625,643
    bool p3 = false;
    int gpus = 1;  // --gpus N: tiles interleaved over HIP devices 0..N-1, RCCL gather at frame end
    for (int k = 1; k < argc; k++) {
        const std::string a = argv[k];
        auto next = [&](const char *what) -> const char * {
            if (k + 1 >= argc) {
                std::fprintf(stderr, "%s needs a value\n", what);
                std::exit(2);
            }
            return argv[++k];
        };
        if (a == "--scene") scene = next("--scene");
        else if (a == "--width") width = std::atoi(next("--width"));
        else if (a == "--height") height = std::atoi(next("--height"));
        else if (a == "--spp") spp = std::atoi(next("--spp"));
        else if (a == "--depth") depth = std::atoi(next("--depth"));
        else if (a == "--assets") assets = next("--assets");
        else if (a == "--p3") p3 = true;
        else if (a == "--gpus") gpus = std::atoi(next("--gpus"));
        else if (a.rfind("--", 0) == 0) {
            std::fprintf(stderr, "unknown option %s\n", a.c_str());
            return 2;
        } else out = a;  // main.cpp:655-659
    }
    try {
        const auto t0 = std::chrono::steady_clock::now();
        const std::string meshes = assets + "/meshes";
        Camera cam;
        if (scene == "sample") cam = sampleScene(meshes);
        else if (scene == "bunny") cam = bunnyScene(meshes);
        else if (scene == "competition") cam = competitionScene(meshes);
        else if (scene == "cones") cam = coneScene(meshes);
        else throw std::runtime_error("unknown scene " + scene);
        Image image(width, height);
        RenderOptions opt;
        opt.spp = spp;
        opt.maxDepth = depth;
        opt.textureDir = assets + "/textures";
        opt.singleFrame = true;  // this program renders one image, like the reference's main()
        for (int g = 0; g < gpus && gpus > 1; g++) opt.devices.push_back(g);
        const auto t1 = std::chrono::steady_clock::now();
        const rt_stats st = render(image, cam, opt);
        const auto t2 = std::chrono::steady_clock::now();
        const double rays = (double)(st.rays_closest + st.rays_shadow);
        std::cout << "It took " << std::chrono::duration<double>(t2 - t0).count() << " seconds to render the image (scene load "
                  << std::chrono::duration<double>(t1 - t0).count() << " s, GPU " << st.ms_total << " ms, "
                  << rays / st.ms_total / 1e3 << " Mrays/s)" << std::endl;
        printIngest(stdout);  // what the reference's one timer lumps together (main.cpp:541)
        if (p3) image.writeImage(out.c_str());
        else image.writeImageBinary(out.c_str());
        clearScene();
    } catch (std::exception &e) {
        std::fprintf(stderr, "raytracer_hip: %s\n", e.what());
        return 1;
    }
    return 0;
}
