/* link-time stand-in so that DT_NEEDED records the unversioned name */
