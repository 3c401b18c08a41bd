"""util/visualizer.py:11-33 of the reference: `save_images` writes every visual as
<name>_<label>.tif (float32 [H,W,C] in [0,255])."""
import ntpath
import os

from . import util


def save_images(output_images_dir, visuals, image_path, subfolder='', iT=None, other_iT=None):
    short_path = ntpath.basename(image_path[0])
    name = os.path.splitext(short_path)[0]
    for label, im_data in visuals.items():
        try:
            im = util.tensor2im(im_data, iT=iT)
        except Exception:
            im = util.tensor2im(im_data, iT=other_iT)
        save_path = os.path.join(output_images_dir, subfolder, '%s_%s.tif' % (name, label))
        util.mkdir(os.path.join(output_images_dir, subfolder))
        util.save_image(im, save_path)
