"""`--dataset_mode infer4rec`: the reference's test-time dataset (data/infer4rec_dataset.py) -- whole
frames served in video order, with the flow files it expects on disk

    <dataroot>/<flowFolder>/<nFolder>/<warp_method>/noisyinputs/<video>/<from>_<to>.tif

created when missing (`--check_data`, data/base_dataset.py:134-249) by the TV-L1 flow of the HIP runtime.

Sample layout (data/infer4rec_dataset.py:176-230), PD = patch_depth, FD = future_patch_depth:
  'n'    [(PD+FD)*4, h, w]  packed raw frames key .. key+PD+FD-1, /(2^bit_depth - 1), then 2x - 1
  'gt'   [PD*3, H, W]       linear-RGB ground truth of frames key .. key+PD-1, same scaling
  'flow' [PD-1+FD, 2, h, w] flows from frame key+PD-1 to its past, then to its future (raw resolution)
  'gt_path', 'n_path'       paths of frame key+PD-1
"""
import os

import numpy as np
import torch

from ..library import (define_transforms, iio_read, iio_write, list_video_files_at_dir, load_image, pathdiff,
                       warpedimagefile)
from ..util.util import mkdir


class infer4recDataset:
    @staticmethod
    def modify_commandline_options(parser, is_train=True):
        parser.add_argument('--frames2load', type=int, default=10)                      # data/base_dataset.py:48
        parser.add_argument('--crop_data', type=str, default=None,
                            help='Crops all output data from this dataset. --crop_data x,y does img[:x,:y].')
        parser.add_argument('--warpeddata', action='store_true', default=False)
        return parser

    def __init__(self, opt):
        self.opt = opt
        self.T, self.iT = define_transforms()
        self.rootdir = opt.dataroot
        if opt.no_predemosaic:
            assert opt.input_nc == 4, "The the input should be 4 channels!!!"
        else:
            assert opt.input_nc == 3, "The the input should be 3 channels!!!"
        self.ftype = opt.bit_depth
        self.gt_paths = os.path.join(self.rootdir, opt.gtFolder if opt.raw_gt else opt.gt_linear_RGB_Folder)
        self.n_paths = os.path.join(self.rootdir, opt.nFolder)
        if not opt.no_warp:
            self._bridge = None                                                          # created on first use (needs the GPU)
            self.w_paths = os.path.join(self.rootdir, opt.wFolder, opt.nFolder, opt.warp_method, 'noisyinputs')
            self.flow_paths = os.path.join(self.rootdir, opt.flowFolder, opt.nFolder, opt.warp_method, 'noisyinputs')
        videos = opt.videos
        if isinstance(videos, str):
            videos = opt.videos = videos.split(',')

        def _dirs(root):
            return sorted(c.path for c in os.scandir(root)
                          if not c.name.startswith('.') and c.is_dir() and (videos is None or c.name in videos))
        self.gt_paths_list, self.noise_paths_list = _dirs(self.gt_paths), _dirs(self.n_paths)
        assert len(self.gt_paths_list) == len(self.noise_paths_list)
        print('%d videos' % len(self.gt_paths_list))
        self.patch_depth = PD = opt.patch_depth
        self.future_patch_depth = FD = opt.future_patch_depth
        self.where = []
        self.videos_noisy_path, self.videos_gt_path, self.videos_w_path, self.videos_flow_path = [], [], [], []
        if not opt.no_warp:
            self.createWarpedInputData(gen_warp=opt.warpeddata)
            self.createFutureWarpedInputData(gen_warp=opt.warpeddata)
        for gt_video_path, n_video_path in zip(self.gt_paths_list, self.noise_paths_list):
            gt_img_paths = list_video_files_at_dir(gt_video_path)
            n_img_paths = list_video_files_at_dir(n_video_path)
            assert len(gt_img_paths) == len(n_img_paths)
            self.where += [k + len(self.videos_gt_path) for k in range(len(gt_img_paths) - PD - FD + 1)]
            self.videos_noisy_path += n_img_paths
            self.videos_gt_path += gt_img_paths
            if not opt.no_warp:
                for p, n_img_path in enumerate(n_img_paths):
                    w_path, f_path = [], []
                    toCode = os.path.splitext(os.path.basename(n_img_path))[0]
                    wfolder = os.path.join(self.w_paths, pathdiff(n_img_path, self.n_paths))
                    ffolder = os.path.join(self.flow_paths, pathdiff(n_img_path, self.n_paths))
                    for z in range(max(p - PD + 1, 0), min(p + FD + 1, len(n_img_paths))):
                        if p == z:
                            w_path.append(n_img_path)
                            continue
                        fromCode = os.path.splitext(os.path.basename(n_img_paths[z]))[0]
                        w_path.append(warpedimagefile(wfolder, fromCode, toCode))
                        f_path.append(warpedimagefile(ffolder, fromCode, toCode))
                    self.videos_w_path.append(w_path)
                    self.videos_flow_path.append(f_path)

    # -- flow files (data/base_dataset.py:134-249) ---------------------------------------------------
    # The reference computes one flow per loop iteration; here the missing pairs of a pass are collected and handed
    # to the device several at a time (rvdd_tvl1flow_batch).  Same files, same contents.
    FLOW_BATCH = 8

    def _bridge_get(self):
        from ..library import CPPbridge
        if self._bridge is None:
            self._bridge = CPPbridge('./build/libBridge.so')
        return self._bridge

    def _files(self, img2_path, from_path):
        toCode = os.path.splitext(os.path.basename(img2_path))[0]
        fromCode = os.path.splitext(os.path.basename(from_path))[0]
        wfolder = os.path.join(self.w_paths, pathdiff(img2_path, self.n_paths))
        ffolder = os.path.join(self.flow_paths, pathdiff(img2_path, self.n_paths))
        return wfolder, ffolder, warpedimagefile(wfolder, fromCode, toCode), warpedimagefile(ffolder, fromCode, toCode)

    def _flush(self, pending, gen_warp):
        """pending: (img2_path, from_path) pairs whose flow file is missing."""
        from ..util.flow_utils import single_warp
        for k in range(0, len(pending), self.FLOW_BATCH):
            chunk = pending[k:k + self.FLOW_BATCH]
            img2 = [iio_read(a).astype(np.float32) for a, _ in chunk]
            img1 = [iio_read(b).astype(np.float32) for _, b in chunk]
            flows = self._bridge_get().TVL1_flow_batch(img2, img1)                  # util/flow_utils.py:144-145, per pair
            for (a, b), i1, flow in zip(chunk, img1, flows):
                wfolder, ffolder, wimagefile, fimagefile = self._files(a, b)
                iio_write(flow.astype(np.float32), fimagefile)
                if gen_warp and not os.path.isfile(wimagefile):
                    iio_write(single_warp(i1, flow).astype(np.float32), wimagefile)

    def _ensure(self, img2_path, from_path, gen_warp, pending):
        from ..util.flow_utils import single_warp
        wfolder, ffolder, wimagefile, fimagefile = self._files(img2_path, from_path)
        mkdir(ffolder)
        if gen_warp:
            mkdir(wfolder)
        if not os.path.isfile(fimagefile):
            if (img2_path, from_path) not in pending:
                pending.append((img2_path, from_path))
        elif gen_warp and not os.path.isfile(wimagefile):
            img1 = iio_read(from_path).astype(np.float32)
            iio_write(single_warp(img1, iio_read(fimagefile).astype(np.float32)).astype(np.float32), wimagefile)

    def createWarpedInputData(self, gen_warp=False):
        if not self.opt.check_data:
            return
        pending = []
        for video2_path in self.noise_paths_list:
            img2_paths = list_video_files_at_dir(video2_path)
            for z in range(len(img2_paths) - self.patch_depth + 1):
                for n in range(self.patch_depth - 1):
                    self._ensure(img2_paths[z + self.patch_depth - 1], img2_paths[z + n], gen_warp, pending)
        self._flush(pending, gen_warp)

    def createFutureWarpedInputData(self, gen_warp=False):
        if (not self.opt.check_data) or self.future_patch_depth == 0:
            return
        pending = []
        for video2_path in self.noise_paths_list:
            img2_paths = list_video_files_at_dir(video2_path)
            for z in range(len(img2_paths) - self.future_patch_depth):
                for n in range(self.future_patch_depth):
                    self._ensure(img2_paths[z], img2_paths[z + n + 1], gen_warp, pending)
        self._flush(pending, gen_warp)

    # -- samples -------------------------------------------------------------------------------------
    def __len__(self):
        return len(self.where)

    def prepare_epoch(self):
        print("nothing to do in prepare_epoch")

    def data_num_channels(self):
        return 3

    def __getitem__(self, index):
        key = self.where[index]
        PD, FD = self.patch_depth, self.future_patch_depth
        gt = np.asarray([load_image(self.videos_gt_path[key + k], self.ftype) for k in range(PD)], dtype=np.float32)
        if not self.opt.no_warp:
            flows = np.asarray([iio_read(path).astype(np.float32) if os.path.isfile(path)
                                else np.zeros(list(gt.shape[1:3]) + [2], dtype=np.float32)
                                for path in self.videos_flow_path[key + PD - 1]], dtype=np.float32)
            flows = torch.from_numpy(np.ascontiguousarray(flows.transpose(0, 3, 1, 2)))
        else:
            flows = []
        noise = np.asarray([load_image(self.videos_noisy_path[key + k], self.ftype) for k in range(PD + FD)],
                           dtype=np.float32)

        def stack(a):                                     # [K,H,W,C] -> [H,W,K*C], frame-major channels
            a = a.transpose(0, 3, 1, 2)
            return a.reshape([a.shape[0] * a.shape[1], a.shape[2], a.shape[3]]).transpose(1, 2, 0)
        gt, noise = stack(gt), self.T(stack(noise))
        if getattr(self.opt, "crop_data", None) is not None:
            x, y = [int(s) for s in self.opt.crop_data.split(',')]
            noise = noise[:, :x, :y]
            gt = gt[:x, :y, :] if self.opt.raw_gt else gt[:2 * x, :2 * y, :]
        return {'gt': self.T(gt), 'n': noise, 'flow': flows,
                'gt_path': self.videos_gt_path[key + PD - 1], 'n_path': self.videos_noisy_path[key + PD - 1]}
