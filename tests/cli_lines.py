"""Command lines of the reference's test.sh, shared by the CPU (resolution) and GPU (bytes) tests."""
# the four command lines of the reference's test.sh that this program's input formats cover, flag for flag
# (test.sh:21-29, :35-43, :49-57, :66-74; only the file names differ: the clips are not in the repository, and the
# .exr is replaced by the half planes read_exr() leaves in memory).  None passes a range flag.
TEST_SH = {
    "yuv444_to_444": "--src_matrix_coeffs 1 --dst_matrix_coeffs 1 --src_transfer_characteristics 1 --dst_transfer_characteristics 1 "
                     "--src_colour_primaries 1 --dst_colour_primaries 1 --src_filename {src}.yuv --dst_filename {dst}.yuv "
                     "--src_pic_width 2560 --src_pic_height 1600 --src_bit_depth 12 --dst_bit_depth 12 "
                     "--src_chroma_format_idc 3 --dst_chroma_format_idc 3 --verbose_level 4 --src_start_frame 0",
    "yuv444_to_420": "--src_matrix_coeffs 1 --dst_matrix_coeffs 1 --src_transfer_characteristics 1 --dst_transfer_characteristics 1 "
                     "--src_colour_primaries 1 --dst_colour_primaries 1 --src_filename {src}.yuv --dst_filename {dst}.yuv "
                     "--src_pic_width 2560 --src_pic_height 1600 --src_bit_depth 12 --dst_bit_depth 12 "
                     "--src_chroma_format_idc 3 --dst_chroma_format_idc 1 --src_start_frame 0 --verbose_level 4",
    "rgb_to_420_10b": "--src_matrix_coeffs 0 --dst_matrix_coeffs 1 --src_transfer_characteristics 1 --dst_transfer_characteristics 1 "
                      "--src_colour_primaries 1 --dst_colour_primaries 1 --src_filename {src}.rgb --dst_filename {dst}.yuv "
                      "--src_pic_width 2560 --src_pic_height 1600 --src_bit_depth 12 --dst_bit_depth 10 "
                      "--src_chroma_format_idc 3 --dst_chroma_format_idc 1 --src_start_frame 0 --verbose_level 4",
    "exr_to_420_10b": "--src_matrix_coeffs 0 --dst_matrix_coeffs 1 --src_transfer_characteristics 8 --dst_transfer_characteristics 1 "
                      "--src_colour_primaries 1 --dst_colour_primaries 1 --src_filename {src}.f16 --dst_filename {dst}.yuv "
                      "--src_pic_width 1920 --src_pic_height 1080 --src_bit_depth 16 --dst_bit_depth 10 "
                      "--src_chroma_format_idc 3 --dst_chroma_format_idc 1 --verbose_level 4 --src_start_frame 0",
}
