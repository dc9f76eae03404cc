/* The one header of the C ABI, seen from the Swift package (swift/Sources/CRayTracerAMD). */
#include "../../../include/raytracer_amd.h"
