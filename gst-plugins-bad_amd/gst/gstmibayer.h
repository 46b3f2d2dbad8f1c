/* Registration entry points of the two elements of plugin `bayer`
 * (reference: GST_ELEMENT_REGISTER_DECLARE in gst/bayer/gstbayerelements.h:30-31). */
#ifndef MI_GST_BAYER_H
#define MI_GST_BAYER_H

#include <gst/gst.h>
#include <gst/base/gstbasetransform.h>

/* Side-by-side build (make side-by-side -> libgstmibayer.so, -DMIBAYER_SIDE_BY_SIDE):
 * the drop-in build registers the reference's own names, so a registry that also
 * holds the stock gst-plugins-bad keeps only one of the two `bayer` plugins
 * (SURVEY.md section 8(b)).  For A/B runs against the stock element in one
 * process this variant registers plugin `mibayer` with the factories
 * `mibayer2rgb` / `mirgb2bayer` and its own GType names; everything else --
 * metadata, templates, properties, behaviour -- is the same code. */
#ifdef MIBAYER_SIDE_BY_SIDE
#define MIBAYER_PLUGIN_NAME mibayer
#define MIBAYER_FACTORY(name) "mi" name
#define MIBAYER_TYPE_NAME(name) "GstMi" name
#else
#define MIBAYER_PLUGIN_NAME bayer
#define MIBAYER_FACTORY(name) name
#define MIBAYER_TYPE_NAME(name) "Gst" name      /* the reference's GType names */
#endif

/* G_DEFINE_TYPE with the registered name as a string (it differs between the two
 * builds while the C identifiers stay the same); parent = GstBaseTransform */
#define MI_DEFINE_ELEMENT_TYPE(TypeName, type_name, registered_name) \
static void type_name##_class_init (TypeName##Class * klass); \
static void type_name##_init (TypeName * self); \
static void \
type_name##_class_intern_init (gpointer klass, gpointer data) \
{ \
  type_name##_class_init ((TypeName##Class *) klass); \
} \
GType \
type_name##_get_type (void) \
{ \
  static gsize type_id = 0; \
  if (g_once_init_enter (&type_id)) { \
    GType t = g_type_register_static_simple (GST_TYPE_BASE_TRANSFORM, \
        g_intern_static_string (registered_name), sizeof (TypeName##Class), \
        type_name##_class_intern_init, sizeof (TypeName), \
        (GInstanceInitFunc) type_name##_init, 0); \
    g_once_init_leave (&type_id, t); \
  } \
  return type_id; \
}

G_BEGIN_DECLS

gboolean gst_bayer2rgb_register (GstPlugin * plugin);
gboolean gst_rgb2bayer_register (GstPlugin * plugin);

G_END_DECLS
#endif
