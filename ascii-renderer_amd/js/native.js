// native.js - loads the N-API addon (ascii-renderer_amd/lib/ascii_rast_napi.node), which binds the
// C ABI of libascii_rast.so.  There is no JavaScript fallback: without the addon or without a GPU the
// first call throws.
import { createRequire } from 'module';
import { fileURLToPath } from 'url';
import path from 'path';

const here = path.dirname(fileURLToPath(import.meta.url));
export const ADDON_PATH = path.join(here, '..', 'lib', 'ascii_rast_napi.node');

let _addon = null;

export function native() {
  if (_addon) return _addon;
  const require = createRequire(import.meta.url);
  try {
    _addon = require(ADDON_PATH);
  } catch (err) {
    throw new Error('HIP addon not available (' + ADDON_PATH + '): ' + err.message +
      ' - build it with `make -C ascii-renderer_amd/csrc`');
  }
  return _addon;
}

// ar_set_option ids (include/ascii_rast.h)
export const AR_OPT_PATH = 1;
export const AR_OPT_POOL_RECORDS = 2;
export const AR_OPT_OVERLAP = 3;
