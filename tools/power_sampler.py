#!/usr/bin/env python
"""r6: side-process sampler of socket power / power cap / shader clock / temperatures (GPU box only; measurement tool, not product).

`python tools/power_sampler.py out.csv [hz] [stopfile]` polls until `stopfile` appears (or 600 s).  Sources, first that answers:
amdsmi's gpu_metrics table (current_socket_power, current_gfxclks per XCD, temperature_hotspot, throttle status), librocm_smi64
through ctypes, the amdgpu hwmon files.  Every row carries time.time() so tools/r6_power.py can cut it by phase."""
import ctypes as C
import glob
import os
import sys
import time

out = sys.argv[1]
hz = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
stop = sys.argv[3] if len(sys.argv) > 3 else out + '.stop'


def src_amdsmi():
    import amdsmi
    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[0]
    cap = None
    try:
        ci = amdsmi.amdsmi_get_power_cap_info(h)
        cap = ci.get('power_cap')
        if isinstance(cap, (int, float)) and cap > 100000:
            cap = cap / 1e6                                  # microwatts on some versions
    except Exception as e:                                   # noqa: BLE001
        sys.stderr.write('power_cap_info: %r\n' % (e,))
    m0 = amdsmi.amdsmi_get_gpu_metrics_info(h)
    sys.stderr.write('amdsmi metrics keys: %s\n' % sorted(m0.keys()))
    sys.stderr.write('amdsmi first sample: %r\n' % {k: m0[k] for k in m0 if any(s in k for s in ('power', 'gfxclk', 'temperature', 'throttle', 'activity', 'uclk', 'energy'))})

    def num(v):
        return v if isinstance(v, (int, float)) else float('nan')

    def sample():
        m = amdsmi.amdsmi_get_gpu_metrics_info(h)
        clk = m.get('current_gfxclks') or [m.get('current_gfxclk')]
        clk = [c for c in clk if isinstance(c, (int, float)) and 0 < c < 60000]
        return dict(power_w=num(m.get('current_socket_power', m.get('average_socket_power'))), cap_w=cap if cap is not None else float('nan'),
                    sclk_mhz=sum(clk) / len(clk) if clk else float('nan'), sclk_max=max(clk) if clk else float('nan'),
                    sclk_min=min(clk) if clk else float('nan'), hotspot_c=num(m.get('temperature_hotspot')), mem_c=num(m.get('temperature_mem')),
                    uclk_mhz=num(m.get('current_uclk')), throttle=num(m.get('throttle_status', m.get('indep_throttle_status'))),
                    gfx_busy=num(m.get('average_gfx_activity')), energy=num(m.get('energy_accumulator')))
    return 'amdsmi', sample


def src_rsmi():
    L = C.CDLL('/opt/rocm/lib/librocm_smi64.so')
    if L.rsmi_init(C.c_uint64(0)) != 0:
        raise RuntimeError('rsmi_init')
    u64 = C.c_uint64

    class Freqs(C.Structure):
        _fields_ = [('has_deep_sleep', C.c_bool), ('num_supported', C.c_uint32), ('current', C.c_uint32), ('frequency', u64 * 33)]

    def sample():
        p, cap, t = u64(0), u64(0), C.c_int64(0)
        f = Freqs()
        if L.rsmi_dev_current_socket_power_get(0, C.byref(p)) != 0:
            L.rsmi_dev_power_ave_get(0, 0, C.byref(p))
        L.rsmi_dev_power_cap_get(0, 0, C.byref(cap))
        sclk = float('nan')
        if L.rsmi_dev_gpu_clk_freq_get(0, 0, C.byref(f)) == 0 and f.current < 33:
            sclk = f.frequency[f.current] / 1e6
        L.rsmi_dev_temp_metric_get(0, 1, 0, C.byref(t))       # junction, current
        return dict(power_w=p.value / 1e6, cap_w=cap.value / 1e6, sclk_mhz=sclk, sclk_max=sclk, sclk_min=sclk, hotspot_c=t.value / 1e3,
                    mem_c=float('nan'), uclk_mhz=float('nan'), throttle=float('nan'), gfx_busy=float('nan'), energy=float('nan'))
    sample()
    return 'rsmi', sample


def src_sysfs():
    hw = glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')
    hw = [h for h in hw if os.path.exists(h + '/power1_average') or os.path.exists(h + '/power1_input')]
    if not hw:
        raise RuntimeError('no hwmon')
    h = hw[0]

    def rd(name, scale):
        try:
            with open(h + '/' + name) as f:
                return int(f.read()) / scale
        except Exception:                                    # noqa: BLE001
            return float('nan')

    def sample():
        p = rd('power1_input', 1e6)
        if p != p:
            p = rd('power1_average', 1e6)
        s = rd('freq1_input', 1e6)
        return dict(power_w=p, cap_w=rd('power1_cap', 1e6), sclk_mhz=s, sclk_max=s, sclk_min=s, hotspot_c=rd('temp2_input', 1e3),
                    mem_c=rd('temp3_input', 1e3), uclk_mhz=rd('freq2_input', 1e6), throttle=float('nan'), gfx_busy=float('nan'), energy=float('nan'))
    sample()
    return 'sysfs', sample


name, sample = None, None
for mk in (src_amdsmi, src_rsmi, src_sysfs):
    try:
        name, sample = mk()
        sample()
        break
    except Exception as e:                                   # noqa: BLE001
        sys.stderr.write('%s unavailable: %r\n' % (mk.__name__, e))
if sample is None:
    sys.stderr.write('no power source on this box\n')
    sys.exit(3)
sys.stderr.write('power source: %s\n' % name)
cols = ['t', 'power_w', 'cap_w', 'sclk_mhz', 'sclk_min', 'sclk_max', 'hotspot_c', 'mem_c', 'uclk_mhz', 'throttle', 'gfx_busy', 'energy']
t_end = time.time() + 600
with open(out, 'w') as f:
    f.write('# source=%s\n' % name)
    f.write(','.join(cols) + '\n')
    nxt = time.time()
    while not os.path.exists(stop) and time.time() < t_end:
        try:
            s = sample()
        except Exception as e:                               # noqa: BLE001
            sys.stderr.write('sample failed: %r\n' % (e,))
            time.sleep(0.2)
            continue
        s['t'] = time.time()
        f.write(','.join('%.6f' % s[c] if c == 't' else '%.3f' % s[c] for c in cols) + '\n')
        f.flush()
        nxt += 1.0 / hz
        d = nxt - time.time()
        if d > 0:
            time.sleep(d)
        else:
            nxt = time.time()
