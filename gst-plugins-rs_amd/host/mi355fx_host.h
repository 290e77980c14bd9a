/* mi355fx_host.h — host-side mirror (C++ implementation, C linkage) of the parts of the reference
 * elements that stay on the CPU: the .cube reader today; the element objects (properties, caps,
 * start/stop, transform vfuncs) in host/elements.*.
 * Reference: video/colorlut/src/parser.rs (CubeLut::parse / parse_file). */
#ifndef MI355FX_HOST_H
#define MI355FX_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi355h_cube mi355h_cube;

/* CubeLut::parse (parser.rs:110-281). NULL on error; `err` receives the CubeParseError text. */
mi355h_cube *mi355h_cube_parse(const char *text, size_t len, char *err, size_t errlen);
/* CubeLut::parse_file (parser.rs:105-108). */
mi355h_cube *mi355h_cube_parse_file(const char *path, char *err, size_t errlen);
void mi355h_cube_free(mi355h_cube *c);
int mi355h_cube_is3d(const mi355h_cube *c);
size_t mi355h_cube_size(const mi355h_cube *c);
/* 3D: Lut3D::as_flat() (size^3 x [r,g,b,1]); 1D: r|g|b planes. Layout == mi355_colorlut_load input. */
const float *mi355h_cube_table(const mi355h_cube *c);
size_t mi355h_cube_table_len(const mi355h_cube *c);
void mi355h_cube_domain(const mi355h_cube *c, float scale[3], float offset[3]);

/* Position::to_cartesian / to_left_handed / to_right_handed (audio/hrtf/src/spatial.rs:40-70); systems numbered as
 * GstHrtfCoordinateSystem: 0 Cartesian, 1 LeftHanded, 2 RightHanded. Returns 0, or -1 on a bad argument. */
int mi355host_position_convert(int from, int to, const float in[3], float out[3]);

/* TryFrom<AudioChannelPosition> for SpatialObject (audio/hrtf/src/spatial.rs:177-222): where hrtfrender puts a positioned input
 * channel when no spatial-objects were set (left-handed coordinates, distance gain 1). `position` is a GstAudioChannelPosition
 * value. Returns 0, or -1 for a position the reference does not support. */
int mi355host_hrtf_object_from_channel_position(int position, float xyz_left_handed[3]);

/* roundedcorners: generate_alpha_mask + draw_rounded_corners (video/videofx/src/border/imp.rs:57-180) into an A8 plane of
 * `stride` x round_up_2(height) bytes: 0xff everywhere for radius 0, otherwise the filled + stroked rounded rectangle drawn by
 * libcairo (dlopen'ed: the library the reference's cairo crate binds). Returns 0, or -1 with the reference's error text. */
int mi355host_rounded_corners_mask(uint8_t *alpha, int width, int height, int stride, unsigned radius, char *err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
