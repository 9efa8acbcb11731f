// ck_stonegeom.h -- host geometry of SfContours.find_stones (ck_stonegeom.cpp); plain C++, no HIP types
#pragma once
#include <cstdint>
#include <vector>

// bitmap (bw x bh, row-major, 0 / 1) of cv2.drawContours(.., [v], 0, 1, thickness=-1) for a polygon of nv integer
// vertices (x0, y0, x1, y1, ...); (bx, by) = the bitmap's origin in the polygon's coordinates
void ck_raster_polygon(const int32_t* v, int nv, int* bx, int* by, int* bw, int* bh, std::vector<uint8_t>& bits);
// cv2.distanceTransform(img, DIST_L2, DIST_MASK_5) * 65536 as integers
void ck_chamfer5(const uint8_t* img, int h, int w, std::vector<int32_t>& dist);
int ck_has_stone_center(const int32_t* dist, int rows, int cols, double radius);
void ck_find_colors(const int16_t* zones, int R, int C, uint8_t* stones, int stride);
// update_grid (stonesfinder.py:888-947) for one intersection zone
void ck_update_grid_host(const int32_t* lines, int k, const int32_t* box, int16_t* slot);
