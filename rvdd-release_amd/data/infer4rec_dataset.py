"""`--dataset_mode infer4rec`: the reference's test-time dataset (data/infer4rec_dataset.py) -- whole frames
served in video order -- with the flow files it expects on disk,

    <dataroot>/<flowFolder>/<nFolder>/<warp_method>/noisyinputs/<video>/<from>_<to>.tif ,

created when missing (`--check_data`, data/base_dataset.py:134-249) by the TV-L1 flow of the HIP runtime.

One sample (data/infer4rec_dataset.py:176-230), PD = patch_depth, FD = future_patch_depth, frame index k:
  'n'    [(PD+FD)*4, h, w]  packed raw frames k .. k+PD+FD-1, divided by 2^bit_depth - 1, then 2x - 1
  'gt'   [PD*3, H, W]       linear-RGB ground truth of frames k .. k+PD-1, same scaling
  'flow' [PD-1+FD, 2, h, w] flows from frame k+PD-1 to its past, then to its future (raw resolution)
  'gt_path', 'n_path'       paths of frame k+PD-1
Public attributes other modules read: `n_paths` (validate.py uses it to name the output folders),
`gt_paths_list`, `noise_paths_list`, `where`, `videos_noisy_path`, `videos_gt_path`, `videos_flow_path`.
"""
import os

import numpy as np
import torch

from ..library import (define_transforms, iio_read, iio_write, list_video_files_at_dir, load_image, pathdiff,
                       warpedimagefile)
from ..util.util import mkdir


def _stem(path):
    return os.path.splitext(os.path.basename(path))[0]


def _video_dirs(root, wanted):
    """Sub-folders of `root` (one per video), hidden ones skipped, optionally restricted to `wanted` names."""
    return sorted(e.path for e in os.scandir(root)
                  if e.is_dir() and not e.name.startswith('.') and (wanted is None or e.name in wanted))


class infer4recDataset:
    FLOW_BATCH = 8                      # missing flows are computed this many at a time (rvdd_tvl1flow_batch)

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
        expected_nc = 4 if opt.no_predemosaic else 3
        assert opt.input_nc == expected_nc, "The the input should be %d channels!!!" % expected_nc
        self.ftype = opt.bit_depth
        self.patch_depth, self.future_patch_depth = opt.patch_depth, opt.future_patch_depth
        self.gt_paths = os.path.join(self.rootdir, opt.gtFolder if opt.raw_gt else opt.gt_linear_RGB_Folder)
        self.n_paths = os.path.join(self.rootdir, opt.nFolder)
        self.warping = not opt.no_warp
        if self.warping:
            self._bridge = None         # the TV-L1 bridge needs the GPU: created on first use
            tail = os.path.join(opt.nFolder, opt.warp_method, 'noisyinputs')
            self.w_paths = os.path.join(self.rootdir, opt.wFolder, tail)
            self.flow_paths = os.path.join(self.rootdir, opt.flowFolder, tail)
        if isinstance(opt.videos, str):
            opt.videos = opt.videos.split(',')
        self.gt_paths_list = _video_dirs(self.gt_paths, opt.videos)
        self.noise_paths_list = _video_dirs(self.n_paths, opt.videos)
        assert len(self.gt_paths_list) == len(self.noise_paths_list)
        print('%d videos' % len(self.gt_paths_list))
        if self.warping:
            self.createWarpedInputData(gen_warp=opt.warpeddata)
            self.createFutureWarpedInputData(gen_warp=opt.warpeddata)
        self._index_frames()

    def _index_frames(self):
        """Flat frame lists over all videos + `where`: the first frame of every sample (no sample spans two videos)."""
        PD, FD = self.patch_depth, self.future_patch_depth
        self.where = []
        self.videos_noisy_path, self.videos_gt_path, self.videos_w_path, self.videos_flow_path = [], [], [], []
        for gt_dir, n_dir in zip(self.gt_paths_list, self.noise_paths_list):
            gt_frames, n_frames = list_video_files_at_dir(gt_dir), list_video_files_at_dir(n_dir)
            assert len(gt_frames) == len(n_frames)
            first = len(self.videos_gt_path)
            self.where.extend(first + k for k in range(len(gt_frames) - PD - FD + 1))
            self.videos_gt_path.extend(gt_frames)
            self.videos_noisy_path.extend(n_frames)
            if not self.warping:
                continue
            for p, target in enumerate(n_frames):          # per frame p: the files that align its neighbours onto it
                sources = [z for z in range(max(p - PD + 1, 0), min(p + FD + 1, len(n_frames)))]
                wfolder, ffolder = self._folders(target)
                self.videos_w_path.append([target if z == p else warpedimagefile(wfolder, _stem(n_frames[z]), _stem(target))
                                           for z in sources])
                self.videos_flow_path.append([warpedimagefile(ffolder, _stem(n_frames[z]), _stem(target))
                                              for z in sources if z != p])

    # -- flow files (data/base_dataset.py:134-249) ---------------------------------------------------
    # The reference computes one flow per loop iteration; here the missing pairs of a pass are collected and handed
    # to the device several at a time.  Same files, same contents.
    def _folders(self, target):
        sub = pathdiff(target, self.n_paths)
        return os.path.join(self.w_paths, sub), os.path.join(self.flow_paths, sub)

    def _bridge_get(self):
        from ..library import CPPbridge
        if self._bridge is None:
            self._bridge = CPPbridge('./build/libBridge.so')
        return self._bridge

    def _queue(self, target, source, gen_warp, pending):
        """Note the pair (flow from `target` to `source`) if its flow file is missing; a missing warped image whose
        flow exists is written right away."""
        from ..util.flow_utils import single_warp
        wfolder, ffolder = self._folders(target)
        mkdir(ffolder)
        if gen_warp:
            mkdir(wfolder)
        ffile = warpedimagefile(ffolder, _stem(source), _stem(target))
        wfile = warpedimagefile(wfolder, _stem(source), _stem(target))
        if not os.path.isfile(ffile):
            if (target, source) not in pending:
                pending.append((target, source))
        elif gen_warp and not os.path.isfile(wfile):
            moved = single_warp(iio_read(source).astype(np.float32), iio_read(ffile).astype(np.float32))
            iio_write(moved.astype(np.float32), wfile)

    def _compute(self, pending, gen_warp):
        from ..util.flow_utils import single_warp
        for k in range(0, len(pending), self.FLOW_BATCH):
            chunk = pending[k:k + self.FLOW_BATCH]
            targets = [iio_read(t).astype(np.float32) for t, _ in chunk]
            sources = [iio_read(s).astype(np.float32) for _, s in chunk]
            flows = self._bridge_get().TVL1_flow_batch(targets, sources)            # util/flow_utils.py:144-145, per pair
            for (t, s), img, flow in zip(chunk, sources, flows):
                wfolder, ffolder = self._folders(t)
                iio_write(flow.astype(np.float32), warpedimagefile(ffolder, _stem(s), _stem(t)))
                wfile = warpedimagefile(wfolder, _stem(s), _stem(t))
                if gen_warp and not os.path.isfile(wfile):
                    iio_write(single_warp(img, flow).astype(np.float32), wfile)

    def createWarpedInputData(self, gen_warp=False):
        """Flows (and optionally warped frames) from every frame to its PD-1 predecessors."""
        if not self.opt.check_data:
            return
        pending, back = [], self.patch_depth - 1
        for n_dir in self.noise_paths_list:
            frames = list_video_files_at_dir(n_dir)
            for p in range(back, len(frames)):
                for z in range(p - back, p):
                    self._queue(frames[p], frames[z], gen_warp, pending)
        self._compute(pending, gen_warp)

    def createFutureWarpedInputData(self, gen_warp=False):
        """... and to its FD successors."""
        FD = self.future_patch_depth
        if (not self.opt.check_data) or FD == 0:
            return
        pending = []
        for n_dir in self.noise_paths_list:
            frames = list_video_files_at_dir(n_dir)
            for p in range(len(frames) - FD):
                for z in range(p + 1, p + FD + 1):
                    self._queue(frames[p], frames[z], gen_warp, pending)
        self._compute(pending, gen_warp)

    # -- samples -------------------------------------------------------------------------------------
    def __len__(self):
        return len(self.where)

    def prepare_epoch(self):
        print("nothing to do in prepare_epoch")

    def data_num_channels(self):
        return 3

    @staticmethod
    def _frames_to_hwc(frames):
        """[K,H,W,C] -> [H,W,K*C] with frame-major channels."""
        k, h, w, c = frames.shape
        return frames.transpose(1, 2, 0, 3).reshape(h, w, k * c)

    def __getitem__(self, index):
        k0 = self.where[index]
        PD, FD = self.patch_depth, self.future_patch_depth
        last = k0 + PD - 1
        gt = np.stack([load_image(self.videos_gt_path[k0 + k], self.ftype) for k in range(PD)]).astype(np.float32)
        noisy = np.stack([load_image(self.videos_noisy_path[k0 + k], self.ftype) for k in range(PD + FD)]).astype(np.float32)
        flows = []
        if self.warping:
            missing = np.zeros(gt.shape[1:3] + (2,), dtype=np.float32)              # as the reference: gt-sized zeros
            fl = np.stack([iio_read(f).astype(np.float32) if os.path.isfile(f) else missing
                           for f in self.videos_flow_path[last]])
            flows = torch.from_numpy(np.ascontiguousarray(fl.transpose(0, 3, 1, 2)))
        gt, noisy = self._frames_to_hwc(gt), self.T(self._frames_to_hwc(noisy))
        crop = getattr(self.opt, "crop_data", None)
        if crop is not None:
            cx, cy = (int(v) for v in crop.split(','))
            noisy = noisy[:, :cx, :cy]
            gt = gt[:cx, :cy, :] if self.opt.raw_gt else gt[:2 * cx, :2 * cy, :]
        return {'gt': self.T(gt), 'n': noisy, 'flow': flows,
                'gt_path': self.videos_gt_path[last], 'n_path': self.videos_noisy_path[last]}
