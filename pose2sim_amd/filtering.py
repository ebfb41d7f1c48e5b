"""Filtering stage: drop-in for the reference's ``filter_all(config_dict)`` (filtering.py:728-830), with the per-column
Python loops replaced by calls into the HIP engine -- every column of the .trc at once.

What runs where:

* outlier rejection (``reject_outliers``, hampel_filter :63-85) ............ p2s_hampel_kernel
* ``butterworth`` (:437-471) ............................................... p2s_butter_kernel
* ``butterworth_on_speed`` (:474-510) ...................................... first difference and running sum by pandas
  (one vectorised call each, with the reference's fillna / cumsum semantics), the filter itself by p2s_butter_kernel
* ``gaussian`` (:513-529) .................................................. p2s_gauss_kernel, weights from scipy's own kernel
* ``median`` (:561-577) .................................................... p2s_median_kernel
* ``one_euro`` (:87-160) ................................................... p2s_one_euro_kernel

* ``kalman`` (:316-434) .................................................... p2s_kalman_kernel -- PARITY UNPINNED: the
  reference takes the filter and the smoother from filterpy, which is not importable where this was built; the kernel
  follows filterpy's published algorithm and is checked against oracle/filtering_ref.py only

Coefficients and kernel weights come from the very SciPy calls the reference makes, so the kernels reproduce its numbers
to rounding.  ``gcv_spline`` and ``loess`` need make_smoothing_spline's private GCV search / statsmodels: they are
refused with NotImplementedError.  The
figures of the reference (``display_figures``, ``save_filt_plots``) are a GUI matter and not produced.
"""
import glob
import logging
import os

import numpy as np

from . import trc as trc_mod

FILTER_HAMPEL, FILTER_GAUSSIAN, FILTER_MEDIAN, FILTER_ONE_EURO, FILTER_KALMAN = 1, 2, 3, 4, 5    # include/p2s.h
REFUSED_TYPES = {'gcv_spline': 'scipy.interpolate.make_smoothing_spline and its GCV search', 'loess': 'statsmodels'}


def _make_engine():
    from .engine import Engine
    return Engine(int(os.environ.get('LOCAL_RANK', '0')))


# ---- the filters, each on a whole [n_frames][n_cols] matrix -------------------------------------------------------------
def butterworth_coefficients(order, cutoff, frame_rate):
    """(b, a, zi) exactly as filtering.py:453-457 builds them and scipy.signal.filtfilt initialises its passes."""
    from scipy import signal
    b, a = signal.butter(int(order) / 2, int(cutoff) / (frame_rate / 2), 'low', analog=False)
    return np.asarray(b, dtype=np.float64), np.asarray(a, dtype=np.float64), np.asarray(signal.lfilter_zi(b, a), dtype=np.float64)


def butterworth_filter(data, order, cutoff, frame_rate, engine=None):
    """butterworth_filter_1d (filtering.py:437-471) on every column."""
    b, a, zi = butterworth_coefficients(order, cutoff, frame_rate)
    engine = engine or _make_engine()
    return engine.butterworth(np.asarray(data, dtype=np.float64), b, a, zi)


def hampel_filter(data, engine=None, n_sigma=2.0):
    """hampel_filter (filtering.py:63-85, window 7) on every column."""
    engine = engine or _make_engine()
    return engine.filter_columns(FILTER_HAMPEL, data, [float(n_sigma)])


def butterworth_on_speed_filter(data, order, cutoff, frame_rate, engine=None):
    """butterworth_on_speed_filter_1d (filtering.py:474-510) on every column: the speed is the first difference (its
    missing values ALL replaced by half the second one, :494), the zero-phase filter runs over it as over positions, and
    the running sum (NaN skipped, :508) starts again from the first position."""
    import pandas as pd
    frame = pd.DataFrame(np.asarray(data, dtype=np.float64))
    speed = frame.diff()
    if len(frame) > 1:
        speed = speed.fillna(speed.iloc[1] / 2)
    filtered = butterworth_filter(speed.to_numpy(), order, cutoff, frame_rate, engine)
    return (pd.DataFrame(filtered).cumsum() + frame.iloc[0]).to_numpy()


def gaussian_filter(data, sigma_kernel, engine=None):
    """gaussian_filter_1d (filtering.py:513-529) on every column."""
    from scipy.ndimage import _filters
    sigma = int(sigma_kernel)
    radius = int(4.0 * float(sigma) + 0.5)                     # gaussian_filter1d's truncate = 4
    weights = np.asarray(_filters._gaussian_kernel1d(sigma, 0, radius)[::-1], dtype=np.float64)
    engine = engine or _make_engine()
    return engine.filter_columns(FILTER_GAUSSIAN, data, weights)


def median_filter(data, kernel_size, engine=None):
    """median_filter_1d (filtering.py:561-577) on every column."""
    engine = engine or _make_engine()
    return engine.filter_columns(FILTER_MEDIAN, data, [float(kernel_size)])


def one_euro_filter(data, frame_rate, min_cutoff=2.5, beta=0.9, d_cutoff=1.0, engine=None):
    """one_euro_filter_1d (filtering.py:87-160) on every column."""
    engine = engine or _make_engine()
    return engine.filter_columns(FILTER_ONE_EURO, data, [1.0 / frame_rate, float(min_cutoff), float(beta), float(d_cutoff)])


def kalman_filter(data, frame_rate, trust_ratio, smooth=True, engine=None):
    """kalman_filter_1d (filtering.py:402-434) on every column: constant-acceleration Kalman filter (measurement noise 20,
    process noise 20 * trust_ratio) and Rauch-Tung-Striebel smoother over every run of >= 4 samples that are neither NaN
    nor 0.  PARITY UNPINNED: the reference takes both from filterpy, which is not importable here -- the kernel follows
    filterpy's published algorithm and is checked against oracle/filtering_ref.py only."""
    engine = engine or _make_engine()
    measurement_noise = 20
    return engine.filter_columns(FILTER_KALMAN, data, [1.0 / frame_rate, float(measurement_noise), float(measurement_noise * int(trust_ratio)),
                                                       1.0 if int(smooth) else 0.0])


def _apply(filter_type, fcfg, data, frame_rate, engine):
    """filter1d (filtering.py:632-662) for a whole matrix."""
    if filter_type == 'butterworth':
        p = fcfg.get('butterworth')
        return butterworth_filter(data, p.get('order'), p.get('cut_off_frequency'), frame_rate, engine)
    if filter_type == 'butterworth_on_speed':
        p = fcfg.get('butterworth_on_speed')
        return butterworth_on_speed_filter(data, p.get('order'), p.get('cut_off_frequency'), frame_rate, engine)
    if filter_type == 'gaussian':
        return gaussian_filter(data, fcfg.get('gaussian').get('sigma_kernel'), engine)
    if filter_type == 'median':
        return median_filter(data, fcfg.get('median').get('kernel_size'), engine)
    if filter_type == 'one_euro':
        p = fcfg.get('one_euro')
        return one_euro_filter(data, frame_rate, p.get('cut_off_frequency', 2.5), p.get('beta', 0.9), p.get('d_cut_off_frequency', 1.0), engine)
    if filter_type == 'kalman':
        p = fcfg.get('kalman')
        return kalman_filter(data, frame_rate, p.get('trust_ratio'), p.get('smooth'), engine)
    if filter_type in REFUSED_TYPES:
        raise NotImplementedError(f"filter type '{filter_type}' needs {REFUSED_TYPES[filter_type]}, which is not part of this build")
    raise KeyError(filter_type)                                # the reference's filter_mapping[filter_type]


# ---- the stage ------------------------------------------------------------------------------------------------------
def _frame_rate(config_dict, project_dir):
    """filtering.py:762-774: the configured rate, or for 'auto' the first video's (60-fps warning text, 30 fps value,
    as in the reference)."""
    frame_rate = config_dict.get('project').get('frame_rate')
    if frame_rate != 'auto':
        return frame_rate
    video_files = glob.glob(os.path.join(project_dir, 'videos', '*' + config_dict['pose']['vid_img_extension']))
    try:
        rate = trc_mod.mp4_frame_rate(video_files[0])
        if not rate:
            raise ValueError
        return round(rate)
    except Exception:
        logging.warning('Cannot read video. Frame rate will be set to 60 fps.')
        return 30


def _sub(fcfg, key, default=None):
    table = fcfg.get(key)
    return table if isinstance(table, dict) else (default or {})


_TYPE_LINES = {
    'butterworth': lambda f: f"--> Filter type: Butterworth low-pass. Order {int(_sub(f, 'butterworth').get('order'))}, Cut-off frequency {int(_sub(f, 'butterworth').get('cut_off_frequency'))} Hz.",
    'one_euro': lambda f: (f"--> Filter type: OneEuro (zero-phase). Min cutoff frequency: {_sub(f, 'one_euro').get('cut_off_frequency', 2.5)} Hz, "
                           f"Beta: {_sub(f, 'one_euro').get('beta', 0.9)}, Derivative cutoff frequency: {_sub(f, 'one_euro').get('d_cut_off_frequency', 1.0)} Hz."),
    'butterworth_on_speed': lambda f: (f"--> Filter type: Butterworth on speed low-pass. Order {int(_sub(f, 'butterworth_on_speed').get('order'))}, "
                                       f"Cut-off frequency {int(_sub(f, 'butterworth_on_speed').get('cut_off_frequency'))} Hz."),
    'gaussian': lambda f: f"--> Filter type: Gaussian. Standard deviation kernel: {int(_sub(f, 'gaussian').get('sigma_kernel'))}",
    'median': lambda f: f"--> Filter type: Median. Kernel size: {_sub(f, 'median').get('kernel_size')}",
    'kalman': lambda f: (f"--> Filter type: Kalman {'smoother' if int(_sub(f, 'kalman').get('smooth')) else 'filter'}. Measurements trusted "
                         f"{int(_sub(f, 'kalman').get('trust_ratio'))} times as much as previous data, assuming a constant acceleration process."),
}


def recap_filter3d(config_dict, trc_path):
    """The stage's report (filtering.py:665-725) for the filter types this build runs."""
    fcfg = config_dict.get('filtering')
    say = logging.info
    say('--> Outliers rejected with a Hampel filter.' if fcfg.get('reject_outliers', False)
        else '--> No outlier rejection applied. Set reject_outliers to true in Config.toml to reject outliers.')
    if fcfg.get('filter', True):
        say(_TYPE_LINES[fcfg.get('type')](fcfg))
    else:
        say('--> No filtering applied. Set filtering to true in Config.toml to filter coordinates.')
    say(f'Filtered 3D coordinates are stored at {trc_path}.')


def _select_frames(frames, frame_range):
    """filtering.py:786-792: the whole file unless frame_range lies inside it; -> (f_range, first row, one past the last)."""
    first, last = int(frames[0]), int(frames[-1])
    whole = frame_range in ('all', 'auto', []) or first > frame_range[0] or int(frames[1]) < frame_range[1]
    f_range = [first, last + 1] if whole else frame_range
    lo = int(np.flatnonzero(frames == f_range[0])[0])
    hi = int(np.flatnonzero(frames == f_range[1] - 1)[0]) + 1
    return f_range, lo, hi


def _patched_header(header, name_in, name_out, frame_nb):
    """filtering.py:795-798: the new file name in line 1, the frame count in fields 3 and 8 of line 3 (the last field
    closes the line)."""
    head = list(header)
    head[0] = head[0].replace(name_in, name_out)
    fields = head[2].split('\t')
    fields[2] = str(frame_nb)
    fields[7] = str(frame_nb) + '\n'
    head[2] = '\t'.join(fields)
    return head


def filter_all(config_dict, engine=None):
    """Same contract as the reference's filter_all: every `pose-3d/*.trc` without 'filt' in its path ->
    `<name>_<f0>-<f1>_filt_<type>.trc` with the coordinates filtered column by column.  Returns the paths written."""
    project_dir = config_dict.get('project').get('project_dir')
    pose3d_dir = os.path.realpath(os.path.join(project_dir, 'pose-3d'))
    fcfg = config_dict.get('filtering')
    do_filter = fcfg.get('filter', True)
    reject_outliers = fcfg.get('reject_outliers', False)
    filter_type = fcfg.get('type')
    frame_range = config_dict.get('project').get('frame_range')
    if do_filter and filter_type in REFUSED_TYPES:
        raise NotImplementedError(f"filter type '{filter_type}' needs {REFUSED_TYPES[filter_type]}, which is not part of this build; "
                                  f"supported: {sorted(_TYPE_LINES)}")
    frame_rate = _frame_rate(config_dict, project_dir)

    written = []
    sources = [p for p in glob.glob(os.path.join(pose3d_dir, '*.trc')) if 'filt' not in p]
    for person_id, path_in in enumerate(sources):
        logging.info(f'\nFiltering 3D coordinates for person {person_id}...')
        frames, times, data, markers, header = trc_mod.load_trc(path_in)
        f_range, lo, hi = _select_frames(frames, frame_range)
        frames, times, data = frames[lo:hi], times[lo:hi], data[lo:hi]
        path_out = path_in.replace(path_in.split('_')[-1], f'{f_range[0]}-{f_range[1]}_filt_{filter_type}.trc')
        head = _patched_header(header, os.path.basename(path_in), os.path.basename(path_out), f_range[1] - f_range[0])

        if not do_filter and not reject_outliers:
            logging.warning(f'reject_outliers and filter have been set to false. No further processing done on {path_in}.\n')
            continue
        if reject_outliers:
            data = hampel_filter(data, engine)
        if not do_filter:
            # the reference goes on to plot and write `Q_filt`, which only the filter assigns (:804-815)
            raise UnboundLocalError("local variable 'Q_filt' referenced before assignment")
        filtered = _apply(filter_type, fcfg, data, frame_rate, engine)
        with open(path_out, 'w') as fh:
            fh.writelines(head)
        trc_mod.write_rows(path_out, frames, times, filtered)
        if fcfg.get('make_c3d'):
            logging.warning('make_c3d: the c3d package is not available in this build; only the .trc file was written.')
        recap_filter3d(config_dict, path_out)
        written.append(path_out)
    return written
