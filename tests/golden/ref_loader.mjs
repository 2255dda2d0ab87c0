// Node 12 loader hook used ONLY by gen_from_reference.mjs in the build container:
// the reference tree has no package.json, so its .js files must be declared ES modules.
export async function getFormat(url, ctx, def) {
  if (url.startsWith('file:///root/reference/') && url.endsWith('.js')) return { format: 'module' };
  return def(url, ctx, def);
}
