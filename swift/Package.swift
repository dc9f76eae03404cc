// swift-tools-version:5.9
// Host side of the HIP path-tracing library for the reference's Swift scene API (Linux Swift toolchain).
// NOT COMPILED IN THIS REPOSITORY'S BUILD: the image that builds and tests the library has no `swift`.
// Everything that computes lives behind the C ABI of include/raytracer_amd.h; this package only flattens a
// Scene into the library's records and drives rt_render.
//
//   make -C raytracer_amd/csrc                      # builds raytracer_amd/lib/libraytracer_amd.so
//   swift build -Xlinker -L../raytracer_amd/lib -Xlinker -rpath -Xlinker $PWD/../raytracer_amd/lib
import PackageDescription

let package = Package(
    name: "RayTracerHIP",
    products: [.library(name: "RayTracerHIP", targets: ["RayTracerHIP"])],
    targets: [
        // The C ABI as a system-library style target: the module map names the one header and the library to link.
        .systemLibrary(name: "CRayTracerAMD", path: "Sources/CRayTracerAMD"),
        .target(name: "RayTracerHIP", dependencies: ["CRayTracerAMD"], path: "Sources/RayTracerHIP"),
    ]
)
